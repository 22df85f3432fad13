"""ctypes binding of librender_mi355x.so (include/render_mi355x.h).  Loading fails loudly:
there is no Python or CPU fallback for the compute path."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# APT_LIB_PATH: another build of the same library (A/B timing of kernel variants); still no fallback
LIB_PATH = os.environ.get("APT_LIB_PATH") or os.path.join(_HERE, "librender_mi355x.so")

APT_OK = 0
APT_MODE_KERNEL, APT_MODE_ORACLE = 0, 1
APT_FLAG_RETIRE = 1
APT_FLAG_RR = 2
APT_FLAG_EMISSION = 4
APT_FLAG_BAND_BUFFERS = 8
APT_FLAG_GRID_SLOTS = 16
APT_ERR_DEVICE = 4
APT_DEV_QUEUE_GUARD, APT_DEV_GRID_TURNS, APT_DEV_LDS_BASE, APT_DEV_GRID_MISMATCH = 1, 2, 4, 8      # bits of the device status word (apt_context_check)

# every symbol include/render_mi355x.h declares
ABI_SYMBOLS = ["apt_default_params", "render_do", "apt_set_default_params", "render_do_ex", "render_frame",
               "apt_gen_rays_device", "apt_decode_color_device", "apt_gen_rays_host", "apt_gen_spheres_host",
               "apt_gen_scene_host", "apt_write_ppm", "apt_abi_version", "apt_last_error", "apt_device_count",
               "apt_set_trace_counter", "apt_selftest_sqrt", "apt_selftest_div3", "apt_set_refill_lanes", "apt_test_scene", "apt_mt19937_checkpoints_host", "apt_gen_rays_mt_device", "apt_build_grid_host",
               "apt_last_status", "apt_render_do", "apt_render_host", "apt_context_create", "apt_context_destroy",
               "apt_context_set_params", "apt_context_set_trace_counter", "apt_context_set_refill_lanes",
               "apt_context_render_do", "apt_context_render_do_ex", "apt_context_render_frame",
               "apt_multi_create", "apt_multi_render", "apt_multi_destroy",
               "apt_decode_color_band", "apt_mt19937_checkpoints_window", "apt_gen_rays_mt_device_ex", "apt_build_grid_device", "apt_render_frame_mt",
               "apt_context_check", "apt_check", "apt_context_set_debug", "apt_set_debug", "apt_context_get_debug", "apt_get_debug", "apt_grid_flags"]
# the reference declares render_do with C++ linkage (src/main.cpp:9-10): the mangled symbol is exported too
CXX_RENDER_DO = "_Z9render_dojPvS_PhS0_S0_"
ABI_VERSION = 3


class AptError(RuntimeError):
    pass


class RenderParams(ctypes.Structure):
    """apt_render_params: run-time form of the reference's compile-time constants
    (src/common.h:4-14, src/render.cpp:141,194-196)."""
    _fields_ = [("struct_size", ctypes.c_uint32), ("width", ctypes.c_uint32), ("height", ctypes.c_uint32),
                ("samples", ctypes.c_uint32), ("depth", ctypes.c_uint32), ("num_spheres", ctypes.c_uint32),
                ("light_index", ctypes.c_int32), ("eps", ctypes.c_float), ("gain", ctypes.c_float),
                ("mode", ctypes.c_uint32), ("flags", ctypes.c_uint32), ("rr_start", ctypes.c_uint32),
                ("path_begin", ctypes.c_uint64), ("path_count", ctypes.c_uint64), ("seed", ctypes.c_uint64),
                ("accel", ctypes.c_uint64)]

    @property
    def num_paths(self):
        return self.width * self.height * 4 * self.samples

    def copy(self, **kw):
        p = RenderParams.from_buffer_copy(bytes(self))
        for k, v in kw.items():
            setattr(p, k, v)
        return p


_lib = None


def lib():
    """The loaded library; raises AptError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AptError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C ascendpathtracing_amd/csrc` (there is no CPU fallback)")
        # One HIP runtime per process: torch's wheel bundles its own libamdhip64 (soname
        # libamdhip64.so.7) and asks for it by file name, so it must be loaded BEFORE this
        # library, which then binds to the already-loaded soname instead of /opt/rocm's copy.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        try:
            h = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # e.g. libamdhip64 not found
            raise AptError(f"cannot load {LIB_PATH}: {e}") from e
        h.apt_last_error.restype = ctypes.c_char_p
        h.render_do.restype = None
        h.apt_default_params.restype = None
        for name in ABI_SYMBOLS + [CXX_RENDER_DO]:
            getattr(h, name)
        h.apt_context_create.restype = ctypes.c_void_p
        h.apt_context_destroy.restype = None
        h.apt_context_render_do.restype = None
        h.apt_render_do.restype = None
        h.apt_multi_destroy.restype = None
        getattr(h, CXX_RENDER_DO).restype = None
        if h.apt_abi_version() != ABI_VERSION:
            raise AptError("librender_mi355x.so ABI version mismatch")
        _lib = h
    return _lib


def build_id():
    """Identifies the build of the library by its SOURCES (sha256 over ascendpathtracing_amd/csrc/*.{h,hip,cpp}, the Makefile and the
    public header, first 16 hex digits): what profiles/summarize.py stamps the recorded PMC traffic with and bench.py compares
    before it reports that traffic for the library it is running."""
    import glob
    import hashlib
    csrc = os.path.join(_HERE, "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.cpp")))
    files += [os.path.join(csrc, "Makefile"), os.path.join(csrc, "apt_exports.map"), os.path.join(os.path.dirname(_HERE), "include", "render_mi355x.h")]
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def check(rc, what):
    if rc != APT_OK:
        raise AptError(f"{what} failed ({rc}): {lib().apt_last_error().decode()}")


def default_params():
    p = RenderParams()
    lib().apt_default_params(ctypes.byref(p))
    return p


def make_params(width=16, height=16, samples=1, depth=5, num_spheres=8, light_index=None, eps=1e-4, gain=12.0,
                mode=APT_MODE_KERNEL, flags=0, path_begin=0, path_count=0, seed=0, rr_start=0, accel=0):
    p = default_params()
    p.width, p.height, p.samples, p.depth = width, height, samples, depth
    p.num_spheres = num_spheres
    p.light_index = num_spheres - 1 if light_index is None else light_index
    p.eps, p.gain, p.mode, p.flags, p.rr_start = eps, gain, mode, flags, rr_start
    p.path_begin, p.path_count, p.seed, p.accel = path_begin, path_count, seed, accel
    return p


def require_gpu():
    if lib().apt_device_count() < 1:
        raise AptError("no HIP device visible: the render path needs an MI355X (no CPU fallback)")
