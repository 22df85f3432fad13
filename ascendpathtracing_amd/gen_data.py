"""Host-side input writers, the counterpart of the reference's scripts/gen_data.py:
gen_rays (:21-75), gen_spheres (:92-132).  The arithmetic runs in librender_mi355x.so
(apt_gen_rays_host / apt_gen_spheres_host / apt_gen_scene_host); outputs are bit-identical
to the reference's rays.bin / spheres.bin for the same (w, h, s, seed)."""
import ctypes
import os

import numpy as np

from ._lib import AptError, check, lib

width, height, samples = 16, 16, 1      # gen_data.py:6-8
eps, bounceMax = 1e-4, 5                # gen_data.py:9-10


def _fptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def gen_rays(w, h, s, seed=0, out_dir=None):
    """-> float32 [6][N] planes ox,oy,oz,dx,dy,dz; writes <out_dir>/rays.bin when given
    (the reference always writes ./input/rays.bin, gen_data.py:71)."""
    n = w * h * 4 * s
    rays = np.empty(6 * n, dtype=np.float32)
    check(lib().apt_gen_rays_host(ctypes.c_uint32(w), ctypes.c_uint32(h), ctypes.c_uint32(s), ctypes.c_uint32(seed),
                                  _fptr(rays)), "apt_gen_rays_host")
    if out_dir is not None:
        rays.tofile(os.path.join(out_dir, "rays.bin"))
    return rays.reshape(6, n)


def mt19937_checkpoints(num_paths, seed=0, stride=64):
    """Host-made MT19937 checkpoints for gen_rays_device(): uint32 [ceil(blocks/stride)][624]."""
    blocks = (num_paths + 155) // 156
    n = (blocks + stride - 1) // stride
    states = np.empty((n, 624), dtype=np.uint32)
    check(lib().apt_mt19937_checkpoints_host(ctypes.c_uint32(seed), ctypes.c_uint64(blocks), ctypes.c_uint32(stride),
                                             states.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))),
          "apt_mt19937_checkpoints_host")
    return states


def mt19937_checkpoints_window(first_block, num_blocks, seed=0, stride=64, state_in=None):
    """Checkpoints of output blocks first_block + k*stride (k < ceil(num_blocks/stride)) -> (uint32 [n][624], raw state
    of block first_block + num_blocks).  state_in = raw state of block first_block (chains windows, or a stored state);
    None walks there from the seed (first_block + 1 sequential twists)."""
    n = (num_blocks + stride - 1) // stride
    states = np.empty((n, 624), dtype=np.uint32)
    out = np.empty(624, dtype=np.uint32)
    st = None if state_in is None else np.ascontiguousarray(state_in, dtype=np.uint32)
    u32p = ctypes.POINTER(ctypes.c_uint32)
    check(lib().apt_mt19937_checkpoints_window(None if st is None else st.ctypes.data_as(u32p), ctypes.c_uint32(seed),
                                               ctypes.c_uint64(first_block), ctypes.c_uint64(num_blocks), ctypes.c_uint32(stride),
                                               states.ctypes.data_as(u32p), out.ctypes.data_as(u32p)),
          "apt_mt19937_checkpoints_window")
    return states, out


def gen_rays_device(w, h, s, seed=0, stride=64, checkpoints=None, stream=None):
    """gen_rays on the GPU, bit-exact with the reference (np.random.seed(seed) MT19937 stream):
    -> torch float32 [6][N] on the device.  `checkpoints` (a CUDA int32/uint32 tensor from
    mt19937_checkpoints) can be passed to reuse a table."""
    import torch
    from . import render
    from ._lib import make_params, require_gpu
    require_gpu()
    p = make_params(w, h, s)
    if checkpoints is None:
        checkpoints = torch.from_numpy(mt19937_checkpoints(p.num_paths, seed, stride).view(np.int32)).cuda()
    rays = torch.empty(6 * p.num_paths, dtype=torch.float32, device="cuda")
    check(lib().apt_gen_rays_mt_device(ctypes.byref(p), render._stream_handle(stream),
                                       ctypes.c_void_p(checkpoints.data_ptr()), ctypes.c_uint32(stride),
                                       ctypes.c_uint64(checkpoints.shape[0]), ctypes.c_void_p(rays.data_ptr())),
          "apt_gen_rays_mt_device")
    return rays.view(6, -1)


def gen_spheres(out_dir=None):
    """-> the 128-float (512-byte) [10][8] table; writes <out_dir>/spheres.bin when given."""
    sph = np.zeros(128, dtype=np.float32)
    check(lib().apt_gen_spheres_host(_fptr(sph)), "apt_gen_spheres_host")
    if out_dir is not None:
        sph.tofile(os.path.join(out_dir, "spheres.bin"))
    return sph


def gen_scene(num_spheres, seed=0, out_dir=None):
    """Build-defined large scene (BASELINE config 4): six walls, Ns-7 random small spheres,
    light at index Ns-1.  -> zero-padded [10][Ns] table."""
    n = ctypes.c_size_t(0)
    check(lib().apt_gen_scene_host(ctypes.c_uint32(num_spheres), ctypes.c_uint64(seed), None, ctypes.byref(n)),
          "apt_gen_scene_host")
    sph = np.zeros(n.value, dtype=np.float32)
    check(lib().apt_gen_scene_host(ctypes.c_uint32(num_spheres), ctypes.c_uint64(seed), _fptr(sph), None),
          "apt_gen_scene_host")
    if out_dir is not None:
        sph.tofile(os.path.join(out_dir, "spheres.bin"))
    return sph


def build_grid(spheres, num_spheres):
    """Host-built uniform grid for a large scene (apt_build_grid_host) -> uint32 numpy buffer to copy to the
    device; pass its device address as RenderParams.accel."""
    spheres = np.ascontiguousarray(spheres, dtype=np.float32).ravel()
    nbytes = ctypes.c_size_t(0)
    check(lib().apt_build_grid_host(_fptr(spheres), ctypes.c_uint32(num_spheres), None, ctypes.byref(nbytes)),
          "apt_build_grid_host")
    buf = np.zeros(nbytes.value // 4, dtype=np.uint32)
    check(lib().apt_build_grid_host(_fptr(spheres), ctypes.c_uint32(num_spheres),
                                    buf.ctypes.data_as(ctypes.c_void_p), ctypes.byref(nbytes)), "apt_build_grid_host")
    return buf


def grid_flags(grid, num_spheres):
    """apt_grid_flags: the flags a built grid earns for a scene of `num_spheres` spheres (APT_FLAG_GRID_SLOTS: one launch per
    frame instead of two) -- `grid` is build_grid()'s numpy buffer or build_grid_device()'s tensor (its 128-byte head is copied
    to the host: a set-up step, not part of a launch path).  OR the result into RenderParams.flags next to accel."""
    # the header is 128 BYTES whatever the element type of the buffer the caller holds it in (apt_grid_flags reads that many)
    if hasattr(grid, "data_ptr"):
        flat = grid.reshape(-1)
        if flat.numel() * flat.element_size() < 128:
            raise AptError("grid_flags: the buffer is shorter than a grid header (128 bytes)")
        n = (128 + flat.element_size() - 1) // flat.element_size()
        head = flat[:n].cpu().contiguous().numpy().view(np.uint8)[:128]
    else:
        flat = np.ascontiguousarray(grid).reshape(-1)
        if flat.nbytes < 128:
            raise AptError("grid_flags: the buffer is shorter than a grid header (128 bytes)")
        head = flat.view(np.uint8)[:128]
    head = np.ascontiguousarray(head)
    lib().apt_grid_flags.restype = ctypes.c_uint32
    return int(lib().apt_grid_flags(head.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(num_spheres)))


def build_grid_device(spheres_dev, num_spheres, stream=None):
    """The same grid built on the GPU from the device-resident table (apt_build_grid_device) -> torch int32 tensor on the
    device (pass its data_ptr() as RenderParams.accel).  Byte-identical to build_grid()."""
    import torch
    from . import render
    from ._lib import require_gpu
    require_gpu()
    st = render._stream_handle(stream)
    nbytes = ctypes.c_size_t(0)
    check(lib().apt_build_grid_device(ctypes.c_void_p(spheres_dev.data_ptr()), ctypes.c_uint32(num_spheres), st, None,
                                      ctypes.c_size_t(0), ctypes.byref(nbytes)), "apt_build_grid_device")
    buf = torch.empty(nbytes.value // 4, dtype=torch.int32, device=spheres_dev.device)
    check(lib().apt_build_grid_device(ctypes.c_void_p(spheres_dev.data_ptr()), ctypes.c_uint32(num_spheres), st,
                                      ctypes.c_void_p(buf.data_ptr()), ctypes.c_size_t(nbytes.value), ctypes.byref(nbytes)),
          "apt_build_grid_device")
    return buf


if __name__ == "__main__":              # gen_data.py:435-446
    os.makedirs("input", exist_ok=True)
    gen_rays(width, height, samples, seed=0, out_dir="./input")
    gen_spheres(out_dir="./input")
    print("===========Python Script Done=============")
