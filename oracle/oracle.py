"""ctypes front end of oracle/libpt_oracle.so -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from
the product package (ascendpathtracing_amd).  See pt_oracle.c for what is restated and the
reference file:line each function follows.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpt_oracle.so")

MODE_K, MODE_O = 0, 1
FLAG_RETIRE = 1
FLAG_RR = 2
FLAG_EMISSION = 4


class Params(ctypes.Structure):
    """Same layout as apt_render_params in include/render_mi355x.h."""
    _fields_ = [("struct_size", ctypes.c_uint32), ("width", ctypes.c_uint32), ("height", ctypes.c_uint32),
                ("samples", ctypes.c_uint32), ("depth", ctypes.c_uint32), ("num_spheres", ctypes.c_uint32),
                ("light_index", ctypes.c_int32), ("eps", ctypes.c_float), ("gain", ctypes.c_float),
                ("mode", ctypes.c_uint32), ("flags", ctypes.c_uint32), ("rr_start", ctypes.c_uint32),
                ("path_begin", ctypes.c_uint64), ("path_count", ctypes.c_uint64), ("seed", ctypes.c_uint64),
                ("accel", ctypes.c_uint64)]


def make_params(width=16, height=16, samples=1, depth=5, num_spheres=8, light_index=None, eps=1e-4, gain=12.0,
                mode=MODE_K, flags=0, path_begin=0, path_count=0, seed=0, rr_start=0, accel=0):
    p = Params()
    p.struct_size = ctypes.sizeof(Params)
    p.width, p.height, p.samples, p.depth = width, height, samples, depth
    p.num_spheres = num_spheres
    p.light_index = num_spheres - 1 if light_index is None else light_index
    p.eps, p.gain, p.mode, p.flags, p.rr_start = eps, gain, mode, flags, rr_start
    p.path_begin, p.path_count, p.seed, p.accel = path_begin, path_count, seed, accel
    return p


def build():
    """Compile libpt_oracle.so (gcc).  Building the checker is not using it."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "pt_oracle.c")):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_scene_floats.restype = ctypes.c_size_t
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, t=ctypes.c_float):
    return a.ctypes.data_as(ctypes.POINTER(t))


def render_paths(params, rays, spheres, threads=1):
    """rays float32 [6][N] (flat ok), spheres [10][Ns] padded -> colors [3][N], traced segments."""
    rays, spheres = _f32(rays).ravel(), _f32(spheres).ravel()
    n = params.width * params.height * 4 * params.samples
    assert rays.size == 6 * n, (rays.size, n)
    colors = np.zeros(3 * n, dtype=np.float32)
    traced = ctypes.c_uint64(0)
    rc = lib().oracle_render_paths(ctypes.byref(params), _ptr(rays), _ptr(spheres), _ptr(colors),
                                   ctypes.c_int(threads), ctypes.byref(traced))
    if rc:
        raise ValueError("oracle_render_paths: bad arguments")
    return colors.reshape(3, n), traced.value


def test_scene(params, rays, spheres):
    rays, spheres = _f32(rays).ravel(), _f32(spheres).ravel()
    n = params.width * params.height * 4 * params.samples
    out = np.zeros(3 * n, dtype=np.float32)
    lib().oracle_test_scene(ctypes.byref(params), _ptr(rays), _ptr(spheres), _ptr(out))
    return out.reshape(3, n)


def gen_rays(w, h, s, seed=0):
    n = w * h * 4 * s
    rays = np.zeros(6 * n, dtype=np.float32)
    lib().oracle_gen_rays(ctypes.c_uint32(w), ctypes.c_uint32(h), ctypes.c_uint32(s), ctypes.c_uint32(seed), _ptr(rays))
    return rays.reshape(6, n)


def gen_rays_window(w, h, s, first_block, first_path, count, seed=0, state_in=None, want_end=False):
    """MT19937 gen_rays for paths [first_path, first_path+count) -> (rays [6][count] band-relative, raw state of block
    first_block uint32[624]) (+ the raw state of block (first_path+count)//156 when want_end: windows chain).
    state_in: the raw state of first_block (skips the walk from the seed)."""
    rays = np.zeros(6 * count, dtype=np.float32)
    st_out = np.zeros(624, dtype=np.uint32)
    st_end = np.zeros(624, dtype=np.uint32)
    st_in = None if state_in is None else np.ascontiguousarray(state_in, dtype=np.uint32)
    rc = lib().oracle_gen_rays_window(ctypes.c_uint32(w), ctypes.c_uint32(h), ctypes.c_uint32(s), ctypes.c_uint32(seed),
                                      None if st_in is None else _ptr(st_in, ctypes.c_uint32), ctypes.c_uint64(first_block),
                                      ctypes.c_uint64(first_path), ctypes.c_uint64(count), _ptr(rays), _ptr(st_out, ctypes.c_uint32),
                                      _ptr(st_end, ctypes.c_uint32) if want_end else None)
    if rc:
        raise ValueError("oracle_gen_rays_window: first_path lies before the window")
    return (rays.reshape(6, count), st_out, st_end) if want_end else (rays.reshape(6, count), st_out)


def gen_rays_counter(params):
    n = params.width * params.height * 4 * params.samples
    rays = np.zeros(6 * n, dtype=np.float32)
    lib().oracle_gen_rays_counter(ctypes.byref(params), _ptr(rays))
    return rays.reshape(6, n)


def path_uniforms(seed, first, n):
    """The counter generator's (u1, u2) of paths [first, first+n) -> float64 [n][2]."""
    out = np.zeros((n, 2), dtype=np.float64)
    lib().oracle_path_uniforms(ctypes.c_uint64(seed), ctypes.c_uint64(first), ctypes.c_uint64(n), _ptr(out, ctypes.c_double))
    return out


def gen_spheres():
    out = np.zeros(128, dtype=np.float32)
    lib().oracle_gen_spheres(_ptr(out))
    return out


def gen_scene(ns, seed=0):
    out = np.zeros(lib().oracle_scene_floats(ctypes.c_uint32(ns)), dtype=np.float32)
    if lib().oracle_gen_scene(ctypes.c_uint32(ns), ctypes.c_uint64(seed), _ptr(out)):
        raise ValueError("oracle_gen_scene: need ns >= 8")
    return out


def decode_color(colors, w, h, s):
    """-> (pre float64 [W*H][3] x-major unflipped, fb float32 [3][W*H], u8 [W*H][3])."""
    colors = _f32(colors).ravel()
    npix = w * h
    assert colors.size == 3 * npix * 4 * s
    pre = np.zeros((npix, 3), dtype=np.float64)
    fb = np.zeros((3, npix), dtype=np.float32)
    u8 = np.zeros((npix, 3), dtype=np.uint8)
    lib().oracle_decode_color(_ptr(colors), ctypes.c_uint32(w), ctypes.c_uint32(h), ctypes.c_uint32(s),
                              _ptr(pre, ctypes.c_double), _ptr(fb), _ptr(u8, ctypes.c_uint8))
    return pre, fb, u8


def write_ppm(path, w, h, u8):
    u8 = np.ascontiguousarray(u8, dtype=np.uint8)
    if lib().oracle_write_ppm(path.encode(), ctypes.c_uint32(w), ctypes.c_uint32(h), _ptr(u8, ctypes.c_uint8)):
        raise OSError("oracle_write_ppm: cannot open " + path)


def render_frame(params, spheres, pixel_begin=0, pixel_count=None, threads=1):
    """-> (fb float32 [3][count], u8 [count][3], pre float64 [count][3], traced)."""
    spheres = _f32(spheres).ravel()
    if pixel_count is None:
        pixel_count = params.width * params.height - pixel_begin
    fb = np.zeros((3, pixel_count), dtype=np.float32)
    u8 = np.zeros((pixel_count, 3), dtype=np.uint8)
    pre = np.zeros((pixel_count, 3), dtype=np.float64)
    traced = ctypes.c_uint64(0)
    rc = lib().oracle_render_frame(ctypes.byref(params), _ptr(spheres), ctypes.c_uint64(pixel_begin),
                                   ctypes.c_uint64(pixel_count), _ptr(fb), _ptr(u8, ctypes.c_uint8),
                                   _ptr(pre, ctypes.c_double), ctypes.c_int(threads), ctypes.byref(traced))
    if rc:
        raise MemoryError("oracle_render_frame")
    return fb, u8, pre, traced.value


def max_threads():
    return lib().oracle_max_threads()
