"""The render boundary, mirroring the reference host interface (src/main.cpp:9-10,74;
src/render.cpp:262-266) on top of the C-ABI.  Tensors are torch CUDA(HIP) tensors used only
as device buffers; the stream is a torch stream (or None = torch's current stream)."""
import ctypes

import torch

from . import _lib
from ._lib import RenderParams, check, lib, require_gpu


def _stream_handle(stream):
    if stream is None:
        stream = torch.cuda.current_stream()
    if isinstance(stream, int):
        return ctypes.c_void_p(stream)
    return ctypes.c_void_p(stream.cuda_stream)


def _dev_f32(t, name, numel=None):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise _lib.AptError(f"{name} must be a contiguous float32 tensor on the GPU")
    if numel is not None and t.numel() != numel:
        raise _lib.AptError(f"{name} has {t.numel()} elements, expected {numel}")
    return ctypes.c_void_p(t.data_ptr())


def _buffer_paths(params):
    """Paths per plane of the ray / colour buffers of a buffer-mode call: the whole image, or the range alone with band-relative buffers."""
    if (params.flags & _lib.APT_FLAG_BAND_BUFFERS) and params.path_count:
        return params.path_count
    return params.num_paths


def sphere_floats(num_spheres):
    """Length of the zero-padded [10][Ns] table (512-byte multiple, gen_data.py:120-127)."""
    return (num_spheres * 10 + 127) // 128 * 128


def render_do(blockDim, l2ctrl, stream, rays, spheres, colors):
    """void render_do(blockDim, l2ctrl, stream, rays, spheres, colors) -- src/main.cpp:9-10,74.
    Asynchronous on `stream`; uses the process-wide defaults (apt_set_default_params)."""
    require_gpu()
    lib().render_do(ctypes.c_uint32(blockDim), None, _stream_handle(stream), _dev_f32(rays, "rays"),
                    _dev_f32(spheres, "spheres"), _dev_f32(colors, "colors"))
    check(lib().apt_last_status(), "render_do")   # void like the reference: the outcome of THIS call, on this thread


class Context:
    """apt_context: private settings for render_do / render_do_ex / render_frame (instead of the process-wide
    default context); safe to use from several threads, one context per thread needs no coordination at all."""

    def __init__(self, params=None):
        self._h = ctypes.c_void_p(lib().apt_context_create())
        if not self._h:
            raise _lib.AptError("apt_context_create failed")
        if params is not None:
            self.set_params(params)

    def close(self):
        if self._h:
            lib().apt_context_destroy(self._h)
            self._h = None

    __del__ = close

    def set_params(self, params):
        check(lib().apt_context_set_params(self._h, ctypes.byref(params)), "apt_context_set_params")

    def set_refill_lanes(self, lanes):
        check(lib().apt_context_set_refill_lanes(self._h, ctypes.c_uint32(lanes)), "apt_context_set_refill_lanes")

    def set_trace_counter(self, tensor_or_none):
        ptr = ctypes.c_void_p(tensor_or_none.data_ptr()) if tensor_or_none is not None else None
        check(lib().apt_context_set_trace_counter(self._h, ptr), "apt_context_set_trace_counter")

    def set_debug(self, key, value):
        check(lib().apt_context_set_debug(self._h, key.encode(), ctypes.c_double(value)), "apt_context_set_debug")

    def get_debug(self, key):
        v = ctypes.c_double(0)
        check(lib().apt_context_get_debug(self._h, key.encode(), ctypes.byref(v)), "apt_context_get_debug")
        return v.value

    def check(self, stream=None):
        """apt_context_check: waits for `stream`, raises AptError when a kernel of this context reported a failure through the
        device status word (the reference asserts inside its kernel, src/render.cpp:68-73)."""
        check(lib().apt_context_check(self._h, _stream_handle(stream)), "apt_context_check")

    def render_do(self, blockDim, l2ctrl, stream, rays, spheres, colors):
        require_gpu()
        lib().apt_context_render_do(self._h, ctypes.c_uint32(blockDim), None, _stream_handle(stream), _dev_f32(rays, "rays"),
                                    _dev_f32(spheres, "spheres"), _dev_f32(colors, "colors"))
        check(lib().apt_last_status(), "apt_context_render_do")

    def render_do_ex(self, params, stream, rays, spheres, colors):
        require_gpu()
        n = _buffer_paths(params)
        check(lib().apt_context_render_do_ex(self._h, ctypes.byref(params), _stream_handle(stream), _dev_f32(rays, "rays", 6 * n),
                                             _dev_f32(spheres, "spheres", sphere_floats(params.num_spheres)),
                                             _dev_f32(colors, "colors", 3 * n)), "apt_context_render_do_ex")

    def render_frame(self, params, spheres, pixel_begin=0, pixel_count=None, stream=None, fb=None, fb_u8=None):
        require_gpu()
        npix = params.width * params.height
        if pixel_count is None:
            pixel_count = npix - pixel_begin
        if fb is None:
            fb = torch.empty((3, pixel_count), dtype=torch.float32, device=spheres.device)
        if fb_u8 is None:
            fb_u8 = torch.empty((pixel_count, 3), dtype=torch.uint8, device=spheres.device)
        check(lib().apt_context_render_frame(self._h, ctypes.byref(params), _stream_handle(stream),
                                             _dev_f32(spheres, "spheres", sphere_floats(params.num_spheres)),
                                             ctypes.c_uint64(pixel_begin), ctypes.c_uint64(pixel_count),
                                             _dev_f32(fb, "fb", 3 * pixel_count), ctypes.c_void_p(fb_u8.data_ptr())),
              "apt_context_render_frame")
        return fb, fb_u8


def render_host(blockDim, rays, spheres, colors):
    """The CPU-simulator shape of the boundary (src/main.cpp:21-44, ICPU_RUN_KF(render, ...)): HOST numpy float32
    buffers in, colours out, synchronous.  Uses the default context's parameters (set_default_params)."""
    import numpy as np
    require_gpu()
    for a, name in ((rays, "rays"), (spheres, "spheres"), (colors, "colors")):
        if not (isinstance(a, np.ndarray) and a.dtype == np.float32 and a.flags.c_contiguous):
            raise _lib.AptError(f"{name} must be a C-contiguous float32 numpy array")
    check(lib().apt_render_host(ctypes.c_uint32(blockDim), ctypes.c_void_p(rays.ctypes.data),
                                ctypes.c_void_p(spheres.ctypes.data), ctypes.c_void_p(colors.ctypes.data)), "apt_render_host")


class MultiGpu:
    """apt_multi: one process, several GPUs.  Band b of the frame renders on device_ids[b]; with stripes > 1 the
    bands are interleaved stripes.  The frame is assembled on device_ids[0] by peer copies (no collective)."""

    def __init__(self, params, spheres_host, device_ids, stripes=1):
        import numpy as np
        require_gpu()
        sph = np.ascontiguousarray(spheres_host, dtype=np.float32)
        if sph.size != sphere_floats(params.num_spheres):
            raise _lib.AptError("spheres_host has the wrong length")
        ids = (ctypes.c_int * len(device_ids))(*device_ids)
        self._h = ctypes.c_void_p()
        check(lib().apt_multi_create(ids, ctypes.c_uint32(len(device_ids)), ctypes.c_uint32(stripes), ctypes.byref(params),
                                     ctypes.c_void_p(sph.ctypes.data), ctypes.byref(self._h)), "apt_multi_create")
        self.params, self.device_ids, self.stripes = params, list(device_ids), stripes
        self.band_kernel_ms = [0.0] * len(device_ids)

    def render(self, fb=None, fb_u8=None):
        """-> (fb [3][W*H] float32, u8 [W*H][3]) on device_ids[0]; synchronous."""
        npix = self.params.width * self.params.height
        dev = torch.device("cuda", self.device_ids[0])
        if fb is None:
            fb = torch.empty((3, npix), dtype=torch.float32, device=dev)
        if fb_u8 is None:
            fb_u8 = torch.empty((npix, 3), dtype=torch.uint8, device=dev)
        ms = (ctypes.c_float * len(self.device_ids))()
        check(lib().apt_multi_render(self._h, ctypes.c_void_p(fb.data_ptr()), ctypes.c_void_p(fb_u8.data_ptr()), ms),
              "apt_multi_render")
        self.band_kernel_ms = list(ms)
        return fb, fb_u8

    def close(self):
        if getattr(self, "_h", None):
            lib().apt_multi_destroy(self._h)
            self._h = None

    __del__ = close


def set_default_params(params):
    check(lib().apt_set_default_params(ctypes.byref(params)), "apt_set_default_params")


def set_debug(key, value):
    """Measurement knob of the default context (include/render_mi355x.h apt_context_set_debug): "queue_ppw", "queue_nbuf",
    "queue_lds_pad", "grid_walk" (1 = nested item walk), "grid_spheres_per_cell"; 0 = the library's own choice."""
    check(lib().apt_set_debug(key.encode(), ctypes.c_double(value)), "apt_set_debug")


def get_debug(key):
    """The knob's current value in the default context (apt_get_debug)."""
    v = ctypes.c_double(0)
    check(lib().apt_get_debug(key.encode(), ctypes.byref(v)), "apt_get_debug")
    return v.value


class debug_knob:
    """with debug_knob("grid_walk", 1): ...  -- sets a knob of the default context and puts back the value it FOUND on the way out
    (an initial value from the environment, or an enclosing debug_knob's), also after an exception (tests select kernels this way,
    never through the process environment)."""

    def __init__(self, key, value):
        self.key, self.value, self.before = key, value, 0

    def __enter__(self):
        self.before = get_debug(self.key)
        set_debug(self.key, self.value)
        return self

    def __exit__(self, *exc):
        set_debug(self.key, self.before)


def check_device_status(stream=None):
    """apt_check: waits for `stream` (None = torch's current stream) and raises AptError when a kernel launched through the default
    context reported a failure through the device status word since the last check (a tripped loop bound: the frame is incomplete)."""
    require_gpu()
    check(lib().apt_check(_stream_handle(stream)), "apt_check")


def render_do_ex(params: RenderParams, stream, rays, spheres, colors):
    """Run-time-parameter form of render_do: rays [6][N], spheres [10][Ns] padded, colors [3][N] (with APT_FLAG_BAND_BUFFERS: planes of
    path_count floats holding only the range)."""
    require_gpu()
    n = _buffer_paths(params)
    check(lib().render_do_ex(ctypes.byref(params), _stream_handle(stream), _dev_f32(rays, "rays", 6 * n),
                             _dev_f32(spheres, "spheres", sphere_floats(params.num_spheres)),
                             _dev_f32(colors, "colors", 3 * n)), "render_do_ex")


def render_paths(params: RenderParams, rays, spheres, stream=None):
    """Convenience: allocate colours, launch, return the [3][N] tensor (not synchronised)."""
    colors = torch.empty(3 * params.num_paths, dtype=torch.float32, device=rays.device)
    render_do_ex(params, stream, rays, spheres, colors)
    return colors.view(3, -1)


def render_frame(params: RenderParams, spheres, pixel_begin=0, pixel_count=None, stream=None, fb=None, fb_u8=None):
    """Fused ray-generate + trace + decode for pixels [pixel_begin, pixel_begin+pixel_count).
    Returns (fb float32 [3][count], fb_u8 uint8 [count][3]); not synchronised."""
    require_gpu()
    npix = params.width * params.height
    if pixel_count is None:
        pixel_count = npix - pixel_begin
    if fb is None:
        fb = torch.empty((3, pixel_count), dtype=torch.float32, device=spheres.device)
    if fb_u8 is None:
        fb_u8 = torch.empty((pixel_count, 3), dtype=torch.uint8, device=spheres.device)
    check(lib().render_frame(ctypes.byref(params), _stream_handle(stream),
                             _dev_f32(spheres, "spheres", sphere_floats(params.num_spheres)),
                             ctypes.c_uint64(pixel_begin), ctypes.c_uint64(pixel_count), _dev_f32(fb, "fb", 3 * pixel_count),
                             ctypes.c_void_p(fb_u8.data_ptr())), "render_frame")
    return fb, fb_u8


def gen_rays_device(params: RenderParams, stream=None, device="cuda"):
    """Device gen_rays with the counter-based generator -> [6][N] tensor."""
    require_gpu()
    rays = torch.empty(6 * params.num_paths, dtype=torch.float32, device=device)
    check(lib().apt_gen_rays_device(ctypes.byref(params), _stream_handle(stream), _dev_f32(rays, "rays")),
          "apt_gen_rays_device")
    return rays.view(6, -1)


def decode_color_device(params: RenderParams, colors, stream=None):
    """Device decode_color: colours [3][N] -> (fb float32 [3][W*H], fb_u8 [W*H][3])."""
    require_gpu()
    npix = params.width * params.height
    fb = torch.empty((3, npix), dtype=torch.float32, device=colors.device)
    u8 = torch.empty((npix, 3), dtype=torch.uint8, device=colors.device)
    check(lib().apt_decode_color_device(ctypes.byref(params), _stream_handle(stream),
                                        _dev_f32(colors.reshape(-1), "colors", 3 * params.num_paths),
                                        ctypes.c_void_p(fb.data_ptr()), ctypes.c_void_p(u8.data_ptr())),
          "apt_decode_color_device")
    return fb, u8


class TraceCounter:
    """Optional device counter of traced ray segments (for reporting under APT_FLAG_RETIRE)."""

    def __init__(self, device="cuda"):
        self.buf = torch.zeros(4, dtype=torch.int64, device=device)  # [0] segments, [1..2] refill statistics

    def __enter__(self):
        self.buf.zero_()
        lib().apt_set_trace_counter(ctypes.c_void_p(self.buf.data_ptr()))
        return self

    def __exit__(self, *exc):
        torch.cuda.synchronize()
        lib().apt_set_trace_counter(None)

    @property
    def value(self):
        return int(self.buf[0].item())

    @property
    def stats(self):
        """(traced segments, lane-slots spent in bounce executions, lane-slots spent in ray-generate) --
        the last two only from the refill loop of render_frame with APT_FLAG_RETIRE; for scenes behind a grid
        they are (cells visited, candidates tested) instead."""
        return tuple(int(x) for x in self.buf[:3].tolist())

    @property
    def exact_reruns(self):
        """Wave-level exact re-runs of a bounce (fast sqrt/divide sequences outside their validity range)."""
        return int(self.buf[3].item())


def selftest_sqrt(variant=0, first_bits=0, count=1 << 32, stream=None):
    """Exhaustive check of the hot loop's fast sqrt against sqrtf(): -> (mismatches, first bad bits)."""
    require_gpu()
    res = torch.tensor([0, -1], dtype=torch.int64, device="cuda")
    check(lib().apt_selftest_sqrt(ctypes.c_int(variant), _stream_handle(stream), ctypes.c_uint64(first_bits),
                                  ctypes.c_uint64(count), ctypes.c_void_p(res.data_ptr())), "apt_selftest_sqrt")
    torch.cuda.synchronize()
    bad, first = res.tolist()
    return bad, first & 0xFFFFFFFF


def selftest_div3(first=0, count=1 << 32, stream=None):
    """Shared-reciprocal divide vs `/` on hashed operand sets: -> (mismatches, first bad, accepted)."""
    require_gpu()
    res = torch.tensor([0, -1, 0], dtype=torch.int64, device="cuda")
    check(lib().apt_selftest_div3(_stream_handle(stream), ctypes.c_uint64(first), ctypes.c_uint64(count),
                                  ctypes.c_void_p(res.data_ptr())), "apt_selftest_div3")
    torch.cuda.synchronize()
    return tuple(res.tolist())


def set_refill_lanes(lanes):
    """Compaction batch threshold (1..64, default 32); speed only, results do not depend on it."""
    check(lib().apt_set_refill_lanes(ctypes.c_uint32(lanes)), "apt_set_refill_lanes")


def test_scene(params: RenderParams, rays, spheres, stream=None):
    """First-hit debug mode (scripts/gen_data.py:134-188 test_scene) -> [3][N] tensor."""
    require_gpu()
    n = params.num_paths
    out = torch.empty(3 * n, dtype=torch.float32, device=rays.device)
    check(lib().apt_test_scene(ctypes.byref(params), _stream_handle(stream), _dev_f32(rays, "rays", 6 * n),
                               _dev_f32(spheres, "spheres", sphere_floats(params.num_spheres)),
                               _dev_f32(out, "out")), "apt_test_scene")
    return out.view(3, -1)


MT_GROUP_PIXELS = 78     # pixels per workgroup of the fused MT19937 frame kernel: 78 * 4 * S paths = 2 S generator blocks


def mt_group_checkpoints(w, h, s, seed=0, pixel_begin=0, pixel_count=None, mt_state=None):
    """Host-made generator states for render_reference_frame_fused(): -> (uint32 [groups][624], first_group).  One raw MT19937
    state per group of 78 pixels (block g * 2 S); mt_state = (block, raw state of that block) starts a far window without walking
    the stream from the seed."""
    import numpy as np
    from . import gen_data
    npix = w * h
    if pixel_count is None:
        pixel_count = npix - pixel_begin
    g_lo, g_hi = pixel_begin // MT_GROUP_PIXELS, (pixel_begin + pixel_count + MT_GROUP_PIXELS - 1) // MT_GROUP_PIXELS
    state = None
    if mt_state is not None:
        blk, raw = mt_state
        if blk > g_lo * 2 * s:
            raise _lib.AptError("mt_state lies after the first block requested")
        state = (np.array(raw, dtype=np.uint32) if blk == g_lo * 2 * s
                 else gen_data.mt19937_checkpoints_window(blk, g_lo * 2 * s - blk, seed, 1 << 30, raw)[1])
    ck, _ = gen_data.mt19937_checkpoints_window(g_lo * 2 * s, (g_hi - g_lo) * 2 * s, seed, 2 * s, state)
    return ck, g_lo


def render_reference_frame_fused(w, h, s, depth=5, seed=0, spheres=None, mode=None, flags=0, stream=None, pixel_begin=0,
                                 pixel_count=None, checkpoints=None, mt_state=None):
    """The reference's whole pipeline (np.random.seed(seed); gen_rays; test_soa arithmetic; decode_color) in ONE launch, with no
    ray / colour buffers in HBM (apt_render_frame_mt): -> (fb float32 [3][pixel_count], fb_u8 [pixel_count][3]), not synchronised;
    bit-identical to render_reference_frame().  Any sample count with a pairwise-sum plan (every s <= 7688; the reference's default is
    16 x 16, s = 1).  checkpoints = (device int32 tensor [groups][624], first_group) from mt_group_checkpoints() to reuse a table."""
    import numpy as np
    from . import gen_data
    from ._lib import APT_MODE_ORACLE, make_params
    require_gpu()
    if spheres is None:
        spheres = torch.from_numpy(gen_data.gen_spheres()).cuda()
    npix = w * h
    if pixel_count is None:
        pixel_count = npix - pixel_begin
    # Everything this call allocates -- the generator states it uploads, the frame buffers -- is allocated and filled ON the stream the
    # kernel runs on (a caller's side stream included), so the upload is ordered before the launch and the caching allocator does not
    # hand the table to another stream while the kernel still reads it (the cross-stream lifetime hazard fixed for
    # render_reference_frame in round 3).
    caller = torch.cuda.current_stream()    # the stream a caller's own table was uploaded on (read BEFORE the `with` below changes it)
    tstream = caller if stream is None else (torch.cuda.ExternalStream(stream) if isinstance(stream, int) else stream)
    with torch.cuda.stream(tstream):
        if checkpoints is None:
            ck, g_lo = mt_group_checkpoints(w, h, s, seed, pixel_begin, pixel_count, mt_state)
            checkpoints = (torch.from_numpy(ck.view(np.int32)).cuda(), g_lo)
        else:
            # a caller's table, made on the caller's current stream: this launch must come after its upload (ordering) and the
            # allocator must not recycle it before the launch is through (lifetime) -- both matter when `stream` is a side stream.
            # (Round 5 waited on torch.cuda.current_stream() INSIDE the `with`: that is tstream itself, a wait on nothing.)
            tstream.wait_stream(caller)
        ck_d, g_lo = checkpoints
        ck_d.record_stream(tstream)
        p = make_params(w, h, s, depth=depth, mode=APT_MODE_ORACLE if mode is None else mode, flags=flags)
        fb = torch.empty((3, pixel_count), dtype=torch.float32, device="cuda")
        u8 = torch.empty((pixel_count, 3), dtype=torch.uint8, device="cuda")
    check(lib().apt_render_frame_mt(ctypes.byref(p), _stream_handle(tstream), ctypes.c_void_p(ck_d.data_ptr()), ctypes.c_uint64(ck_d.shape[0]),
                                    ctypes.c_uint64(g_lo), _dev_f32(spheres, "spheres"), ctypes.c_uint64(pixel_begin),
                                    ctypes.c_uint64(pixel_count), ctypes.c_void_p(fb.data_ptr()), ctypes.c_void_p(u8.data_ptr())),
          "apt_render_frame_mt")
    return fb, u8


def render_reference_frame(w, h, s, depth=5, seed=0, spheres=None, mode=None, flags=0, stream=None, band_pixels=None,
                           pixel_begin=0, pixel_count=None, mt_state=None):
    """The reference's whole pipeline on the device, bit-exact with running scripts/gen_data.py
    (np.random.seed(seed); gen_rays; gen_spheres; test_soa) and scripts/data_visualization.py:
    MT19937 gen_rays -> render (O-mode by default = the NumPy oracle's arithmetic) -> decode_color.

    band_pixels=None: three whole-frame launches with [6][N] / [3][N] intermediates (36 bytes per path: 19 GB at C2)
        -> (fb float32 [3][W*H], fb_u8 uint8 [W*H][3], colors [3][N]); not synchronised.
    band_pixels=B: the same pipeline band by band in band-relative buffers (APT_FLAG_BAND_BUFFERS) of B pixels
        (36*4*S*B bytes, e.g. 604 MB for B = 65536 at S = 64), for pixels [pixel_begin, pixel_begin+pixel_count)
        -> (fb [3][pixel_count], fb_u8 [pixel_count][3], None).  The MT19937 checkpoints are made window by
        window on the host (chained); mt_state = (block, raw uint32[624] state of that block) lets a far window start
        without walking the stream from the seed (C3's last band is 1.1e8 blocks in)."""
    import numpy as np
    from . import gen_data
    from ._lib import APT_FLAG_BAND_BUFFERS, APT_MODE_ORACLE, make_params
    require_gpu()
    if spheres is None:
        spheres = torch.from_numpy(gen_data.gen_spheres()).cuda()
    mode = APT_MODE_ORACLE if mode is None else mode
    if band_pixels is None:
        p = make_params(w, h, s, depth=depth, mode=mode, flags=flags)
        rays = gen_data.gen_rays_device(w, h, s, seed=seed, stream=stream)
        colors = render_paths(p, rays.reshape(-1), spheres, stream=stream)
        fb, u8 = decode_color_device(p, colors, stream=stream)
        return fb, u8, colors
    npix = w * h
    if pixel_count is None:
        pixel_count = npix - pixel_begin
    per_pixel, stride = 4 * s, 64
    fb = torch.empty((3, pixel_count), dtype=torch.float32, device="cuda")
    u8 = torch.empty((pixel_count, 3), dtype=torch.uint8, device="cuda")
    rays = torch.empty(6 * band_pixels * per_pixel, dtype=torch.float32, device="cuda")
    colors = torch.empty(3 * band_pixels * per_pixel, dtype=torch.float32, device="cuda")
    # Everything of a band -- the three launches, the copy of its pixels into `fb` and the wait before its buffers are reused --
    # runs on ONE stream: torch's current stream inside the `with` below (a caller's side stream included; before round 3 the copy
    # and the wait used the current stream while the launches used `stream`).
    tstream = torch.cuda.current_stream() if stream is None else (torch.cuda.ExternalStream(stream) if isinstance(stream, int) else stream)
    st = _stream_handle(tstream)
    state_blk, state = (None, None) if mt_state is None else mt_state

    def advance(raw, blk_from, blk_to):     # raw state of block blk_from -> raw state of block blk_to (host, sequential)
        if blk_to == blk_from:
            return np.array(raw, dtype=np.uint32)
        return gen_data.mt19937_checkpoints_window(blk_from, blk_to - blk_from, seed, 1 << 30, raw)[1]

    for q0 in range(pixel_begin, pixel_begin + pixel_count, band_pixels):
        nq = min(band_pixels, pixel_begin + pixel_count - q0)
        b, c = q0 * per_pixel, nq * per_pixel
        blk0, blk1 = b // 156, (b + c + 155) // 156
        if state_blk is not None and state_blk > blk0:
            raise _lib.AptError("mt_state lies after the first path requested")
        if state_blk is not None and state_blk < blk0:      # walk the stored state forward to this band (host, sequential)
            state, state_blk = advance(state, state_blk, blk0), blk0
        ck, nxt = gen_data.mt19937_checkpoints_window(blk0, blk1 - blk0, seed, stride, state if state_blk == blk0 else None)
        # the window's end state is the raw state of block blk1; the next band starts at block (b + c) // 156 = blk1 or blk1 - 1
        # (a band boundary inside a block), so keep the state of the LAST block of this window instead when they overlap
        with torch.cuda.stream(tstream):                    # the upload is ordered before the launch that reads it
            ck_d = torch.from_numpy(ck.view(np.int32)).cuda()
            fb_band = torch.empty((3, nq), dtype=torch.float32, device="cuda")
        p = make_params(w, h, s, depth=depth, mode=mode, flags=flags | APT_FLAG_BAND_BUFFERS, path_begin=b, path_count=c)
        check(lib().apt_gen_rays_mt_device_ex(ctypes.byref(p), st, ctypes.c_void_p(ck_d.data_ptr()), ctypes.c_uint32(stride),
                                              ctypes.c_uint64(ck.shape[0]), ctypes.c_uint64(blk0), ctypes.c_void_p(rays.data_ptr())),
              "apt_gen_rays_mt_device_ex")
        check(lib().render_do_ex(ctypes.byref(p), st, ctypes.c_void_p(rays.data_ptr()), _dev_f32(spheres, "spheres"),
                                 ctypes.c_void_p(colors.data_ptr())), "render_do_ex")
        o = q0 - pixel_begin
        check(lib().apt_decode_color_band(ctypes.byref(p), st, ctypes.c_void_p(colors.data_ptr()), ctypes.c_uint64(nq),
                                          ctypes.c_void_p(fb_band.data_ptr()), ctypes.c_void_p(u8[o:o + nq].data_ptr())),
              "apt_decode_color_band")
        with torch.cuda.stream(tstream):
            fb[:, o:o + nq] = fb_band
        tstream.synchronize()                               # the checkpoint table of this band may now be freed / rays reused
        state_blk, state = blk1, nxt
        if (b + c) % 156:                                    # the next band starts inside block blk1 - 1: re-derive from this window's table
            last_cp = (blk1 - 1 - blk0) // stride
            state, state_blk = advance(ck[last_cp], blk0 + last_cp * stride, blk1 - 1), blk1 - 1
    return fb, u8, None
