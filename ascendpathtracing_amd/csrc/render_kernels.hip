// render_kernels.hip -- the one translation unit of the device side of librender_mi355x.so: host-side
// launch logic and the C-ABI (include/render_mi355x.h) on top of
//     pt_core.h     the reference's arithmetic, shared with the host helpers
//     pt_trace.h    scene access + bounce loops (8-sphere SGPR path, LDS tiles, grid walk)
//     pt_kernels.h  the __global__ kernels
// One lane = one path; the whole bounce loop lives in registers (the reference moves 64-ray tiles through
// a 16 KB scratch buffer with a free-list allocator, src/allocator.h -- no counterpart here).  No MFMA:
// this is branchy fp32 VALU work whose separately rounded mul/add an MFMA chain could not reproduce.
// Built with -ffp-contract=off -fno-slp-vectorize (Makefile); see DESIGN.md sections 2 and 4.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <new>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/render_mi355x.h"
#include "apt_host.h"
#include "pt_core.h"

#include "pt_kernels.h"
#include "pt_grid_build.h"

namespace {

// ---- host side ----------------------------------------------------------------------------
using apt::clear_error;
int fail(int code, const char *fmt, const char *detail = "") { return apt::set_error(code, fmt, detail); }
int hip_fail(hipError_t e) { return fail(APT_ERR_DEVICE, "HIP: %s", hipGetErrorString(e)); }

int make_leaf_prog(uint32_t samples, LeafProg &lp) {   // the plan itself: pt_leaf.h
    return make_leaf_plan(samples, lp) ? APT_OK : fail(APT_ERR_ARG, "samples too large: its pairwise-sum plan needs more than 64 leaves (every count <= 7688 fits, and 8192)%s");
}

int check_params(const apt_render_params *p) {
    if (!p) return fail(APT_ERR_ARG, "params is null%s");
    if (p->struct_size != sizeof(apt_render_params)) return fail(APT_ERR_STRUCT, "apt_render_params.struct_size mismatch%s");
    if (!p->width || !p->height || !p->samples) return fail(APT_ERR_ARG, "width/height/samples must be non-zero%s");
    if (p->mode > APT_MODE_ORACLE) return fail(APT_ERR_ARG, "unknown mode%s");
    if (p->num_spheres == 0) return fail(APT_ERR_SCENE, "num_spheres is 0%s");
    if (p->light_index >= (int32_t)p->num_spheres) return fail(APT_ERR_SCENE, "light_index out of range%s");
    if ((p->flags & APT_FLAG_EMISSION) && p->light_index < 0) return fail(APT_ERR_SCENE, "APT_FLAG_EMISSION needs a light_index >= 0%s");
    return APT_OK;
}

// The device status word of `ctx` on the current device (include/render_mi355x.h "Device-side failure channel"), made on the
// context's first launch there: 4 bytes of device memory, once.  Null -- the kernels then report nothing -- when it does not exist yet
// and cannot be made now (the stream is being captured, or the allocation failed).
uint32_t *status_word(apt_context &ctx, hipStream_t st) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (dev < 0 || dev >= apt::kMaxStatusDevices) return nullptr;   // beyond the table: no word (and no allocation per launch)
    if (uint32_t *w = ctx.status_lookup(dev)) return w;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (cs != hipStreamCaptureStatusNone) return nullptr;
    uint32_t *fresh = nullptr, *spare = nullptr;
    if (hipMalloc(&fresh, sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipMemset(fresh, 0, sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(fresh); return nullptr; }
    uint32_t *w = ctx.status_adopt(dev, fresh, &spare);
    if (spare) (void)hipFree(spare);      // another thread was first (or the device index is beyond the table)
    return w;
}

// What a render call needs from its context: one consistent copy of the values and the status word of the current device.
struct Launch {
    apt_context::Values cv;
    uint32_t *status;
};
Launch launch_state(apt_context &ctx, void *stream) { return Launch{ctx.snapshot(), status_word(ctx, (hipStream_t)stream)}; }

TraceArgs make_trace_args(const apt_render_params *p, const Launch &ls) {
    const apt_context::Values &cv = ls.cv;
    TraceArgs ta;
    ta.ns = p->num_spheres; ta.depth = p->depth; ta.light = p->light_index;
    ta.eps = p->eps; ta.gain = p->gain; ta.traced = cv.trace_counter;
    ta.status = ls.status;
    ta.refill_lanes = cv.refill_lanes;
    ta.grid = (p->num_spheres != 8) ? reinterpret_cast<const uint32_t *>((uintptr_t)p->accel) : nullptr;
    ta.grid_walk = 0;
    ta.emission = (p->flags & APT_FLAG_EMISSION) ? 1u : 0u;
    ta.rr_start = (p->flags & APT_FLAG_RR) ? (p->rr_start ? p->rr_start : 3u) : 0u;
    ta.seed = p->seed;
    return ta;
}

// Buffer mode takes the two-paths-per-lane kernel from this many paths on: below it the one-path kernel's twice as many workgroups fill the
// chip better (C1, the reference's own CPU-runnable configuration, is 262 144 paths = 1024 workgroups of it: launch-bound either way).
constexpr uint64_t kTwoPathBufferMin = 1ull << 20;

template <int MODE, int SC>
void launch_paths(bool retire, dim3 grid, hipStream_t st, const float *rays, const float *sph, float *colors,
                  uint64_t n, uint64_t b, uint64_t c, const TraceArgs &ta) {
    if (retire) hipLaunchKernelGGL((render_paths_kernel<MODE, SC, true>), grid, dim3(kBlock), 0, st, rays, sph, colors, n, b, c, ta);
    else hipLaunchKernelGGL((render_paths_kernel<MODE, SC, false>), grid, dim3(kBlock), 0, st, rays, sph, colors, n, b, c, ta);
}

template <int MODE, int SC, int GROUP>
void launch_frame(bool retire, dim3 grid, size_t lds, hipStream_t st, const float *sph, const FrameArgs &fa,
                  const TraceArgs &ta, const LeafProg &lp) {
    if constexpr (SC == kScene8 && GROUP == 8) {
        // With APT_FLAG_RETIRE these frames belong to the sample-queue kernel (pt_queue.h); what arrives here with the flag set is
        // depth 0, where there is nothing to retire.  (Round 2's wave queue inside render_frame_kernel for this case is gone.)
        hipLaunchKernelGGL((render_frame_kernel<MODE, SC, GROUP, false>), grid, dim3(kBlock), lds, st, sph, fa, ta, lp);
    } else {
        if (retire) hipLaunchKernelGGL((render_frame_kernel<MODE, SC, GROUP, true>), grid, dim3(kBlock), lds, st, sph, fa, ta, lp);
        else hipLaunchKernelGGL((render_frame_kernel<MODE, SC, GROUP, false>), grid, dim3(kBlock), lds, st, sph, fa, ta, lp);
    }
}

template <int MODE, int SC>
void launch_frame_g(int group, bool retire, dim3 grid, size_t lds, hipStream_t st, const float *sph,
                    const FrameArgs &fa, const TraceArgs &ta, const LeafProg &lp) {
    // the headline case -- 8 spheres, >= 16 samples, every segment traced, no roulette -- runs two paths per lane
    if (SC == kScene8 && group == 8 && !retire && ta.rr_start == 0 && fa.samples >= 16) {
        hipLaunchKernelGGL((render_frame_kernel<MODE, kScene8, 8, false, true>), grid, dim3(kBlock), lds, st, sph, fa, ta, lp);
        return;
    }
    if (group == 8) launch_frame<MODE, SC, 8>(retire, grid, lds, st, sph, fa, ta, lp);
    else launch_frame<MODE, SC, 1>(retire, grid, lds, st, sph, fa, ta, lp);
}

// ---- the two render launches, on an explicit snapshot of a context's values -----------------
int do_render_paths(const Launch &ls, const apt_render_params *p, void *stream, const float *rays,
                    const float *spheres, float *colors) {
    int rc = check_params(p);
    if (rc) return rc;
    if (!rays || !spheres || !colors) return fail(APT_ERR_ARG, "rays/spheres/colors must be non-null%s");
    const uint64_t n_image = (uint64_t)p->width * p->height * 4u * p->samples;
    const uint64_t b = p->path_begin;
    if (b > n_image) return fail(APT_ERR_ARG, "path_begin beyond the image%s");
    const uint64_t c = p->path_count ? p->path_count : n_image - b;
    if (b + c > n_image) return fail(APT_ERR_ARG, "path range beyond the image%s");
    if (c == 0) return APT_OK;
    const uint64_t blocks = (c + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return fail(APT_ERR_ARG, "path_count too large for one launch; shard it%s");
    hipStream_t st = (hipStream_t)stream;
    const bool ns8 = p->num_spheres == 8;
    const TraceArgs ta = make_trace_args(p, ls);
    const bool retire = p->flags & APT_FLAG_RETIRE;
    const dim3 grid((unsigned)blocks);
    const int sck = ns8 ? kScene8 : (ta.grid ? kSceneGrid : kSceneTiles);
    uint64_t n = n_image;
    if (p->flags & APT_FLAG_BAND_BUFFERS) { // planes of c floats holding paths [b, b+c): same indexing through a shifted base
        n = c;
        rays -= b;
        colors -= b;
    }
    if (ns8 && !retire && ta.rr_start == 0 && c >= kTwoPathBufferMin) {   // a large range of the reference scene, every segment traced: two paths per lane
        const uint64_t per_block = 2ull * kBlock * kPaths2Pairs;
        const dim3 grid2((unsigned)((c + per_block - 1) / per_block));
        if (p->mode == APT_MODE_ORACLE) hipLaunchKernelGGL((render_paths2_kernel<kModeOracle>), grid2, dim3(kBlock), 0, st, rays, spheres, colors, n, b, c, ta);
        else hipLaunchKernelGGL((render_paths2_kernel<kModeKernel>), grid2, dim3(kBlock), 0, st, rays, spheres, colors, n, b, c, ta);
    } else if (retire && ns8) { // wave-level queue: one wave per kQueueChunk consecutive paths
        const uint64_t waves = (c + kQueueChunk - 1) / kQueueChunk;
        const dim3 qgrid((unsigned)((waves + kBlock / 64 - 1) / (kBlock / 64)));
        if (p->mode == APT_MODE_ORACLE) hipLaunchKernelGGL((render_paths_queue_kernel<kModeOracle>), qgrid, dim3(kBlock), 0, st, rays, spheres, colors, n, b, c, ta);
        else hipLaunchKernelGGL((render_paths_queue_kernel<kModeKernel>), qgrid, dim3(kBlock), 0, st, rays, spheres, colors, n, b, c, ta);
    } else if (p->mode == APT_MODE_ORACLE) {
        if (sck == kScene8) launch_paths<kModeOracle, kScene8>(retire, grid, st, rays, spheres, colors, n, b, c, ta);
        else if (sck == kSceneGrid) launch_paths<kModeOracle, kSceneGrid>(retire, grid, st, rays, spheres, colors, n, b, c, ta);
        else launch_paths<kModeOracle, kSceneTiles>(retire, grid, st, rays, spheres, colors, n, b, c, ta);
    } else {
        if (sck == kScene8) launch_paths<kModeKernel, kScene8>(retire, grid, st, rays, spheres, colors, n, b, c, ta);
        else if (sck == kSceneGrid) launch_paths<kModeKernel, kSceneGrid>(retire, grid, st, rays, spheres, colors, n, b, c, ta);
        else launch_paths<kModeKernel, kSceneTiles>(retire, grid, st, rays, spheres, colors, n, b, c, ta);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? APT_OK : hip_fail(e);
}

// The sample-queue kernels' launch shape (pt_queue.h): colour buffers, pixels per wave, dynamic LDS.  -> false: too many waves.
bool queue_launch_shape(const apt::Debug &dbg, const LeafProg &lp, bool rr, bool grid, uint64_t pixel_count, bool retire, QueueArgs &qa,
                        uint64_t &waves, size_t &qlds) {
    // a unit = one pairwise leaf of a pixel (8-sphere form) or of half a pixel (grid form); buffers for ~512 resp. ~384 items in flight
    const uint32_t subs = queue_unit_subs(grid), unit_items = subs * lp.maxleaf, window = subs == 4u ? 512u : 384u;
    qa.nbuf = dbg.queue_nbuf ? dbg.queue_nbuf : std::max(2u, std::min(16u, (window + unit_items - 1u) / unit_items));
    qa.buf_bytes = queue_buf_bytes(lp.maxleaf, subs);
    qa.retire = retire ? 1u : 0u;
    // pixels per wave: 16 at C2 (lane efficiency 0.98; 8 / 16 / 24 measured within 0.5 % of each other, 2 costs 6 %), fewer only
    // for frames too small to fill the chip's ~3800 wave slots a few times over
    const uint64_t ppw = dbg.queue_ppw ? dbg.queue_ppw : std::max<uint64_t>(4, std::min<uint64_t>(16, pixel_count / 8192u));
    qa.ppw = (uint32_t)ppw;
    waves = (pixel_count + ppw - 1) / ppw;
    qlds = queue_lds_bytes(queue_pool_entries(grid), rr, qa.nbuf, lp.nleaves > 1, qa.buf_bytes, subs) + dbg.queue_lds_pad;   // the pad: experiments only, lowers the occupancy
    return waves <= 0x7fffffffull;
}

int do_render_frame(const Launch &ls, const apt_render_params *p, void *stream, const float *spheres,
                    uint64_t pixel_begin, uint64_t pixel_count, float *fb, uint8_t *fb_u8) {
    const apt::Debug &dbg = ls.cv.debug;
    int rc = check_params(p);
    if (rc) return rc;
    if (!spheres || !fb) return fail(APT_ERR_ARG, "spheres/fb must be non-null%s");
    const uint64_t npix = (uint64_t)p->width * p->height;
    if (pixel_begin > npix || pixel_count > npix - pixel_begin) return fail(APT_ERR_ARG, "pixel range beyond the image%s");
    if (pixel_count == 0) return APT_OK;
    LeafProg lp;
    if ((rc = make_leaf_prog(p->samples, lp))) return rc;
    const int group = p->samples >= 8 ? 8 : 1;
    const uint64_t lanes = pixel_count * 4u * (uint64_t)group;
    const uint64_t blocks = (lanes + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return fail(APT_ERR_ARG, "pixel_count too large for one launch; shard it%s");
    hipStream_t st = (hipStream_t)stream;
    const bool ns8 = p->num_spheres == 8;
    const TraceArgs ta = make_trace_args(p, ls);
    FrameArgs fa;
    camera_init(fa.cam, p->width, p->height);
    fa.width = p->width; fa.height = p->height; fa.samples = p->samples; fa.seed = p->seed;
    fa.pixel_begin = pixel_begin; fa.pixel_count = pixel_count; fa.fb = fb; fa.fb_u8 = fb_u8;
    const bool retire = p->flags & APT_FLAG_RETIRE;
    if (retire && ns8 && group == 8 && p->depth > 0) {
        // 8-sphere scene with compaction: one wave per workgroup, a stream of `ppw` pixels per wave (pt_queue.h)
        QueueArgs qa;
        uint64_t waves;
        size_t qlds;
        const bool rrk = ta.rr_start != 0;
        if (!queue_launch_shape(dbg, lp, rrk, false, pixel_count, true, qa, waves, qlds)) return fail(APT_ERR_ARG, "pixel_count too large for one launch; shard it%s");
        if (p->mode == APT_MODE_ORACLE) {
            if (rrk) hipLaunchKernelGGL((render_frame_queue8_kernel<kModeOracle, true>), dim3((unsigned)waves), dim3(64), qlds, st, spheres, fa, ta, lp, qa);
            else hipLaunchKernelGGL((render_frame_queue8_kernel<kModeOracle, false>), dim3((unsigned)waves), dim3(64), qlds, st, spheres, fa, ta, lp, qa);
        } else {
            if (rrk) hipLaunchKernelGGL((render_frame_queue8_kernel<kModeKernel, true>), dim3((unsigned)waves), dim3(64), qlds, st, spheres, fa, ta, lp, qa);
            else hipLaunchKernelGGL((render_frame_queue8_kernel<kModeKernel, false>), dim3((unsigned)waves), dim3(64), qlds, st, spheres, fa, ta, lp, qa);
        }
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? APT_OK : hip_fail(e);
    }
    TraceArgs ta_frame = ta;
    if (!ns8 && ta.grid && group == 8 && p->depth > 0 && dbg.grid_walk != 1u) {
        // A scene behind a grid: the sample-queue kernel's grid form (pt_queue.h run_grid), with or without APT_FLAG_RETIRE.  Whether
        // the grid carries the pair-slot tables that form needs is written in the buffer on the DEVICE: rather than reading it back in
        // the launch path, both kernels are launched and each asks grid_queue_usable() -- the one not chosen returns at once.
        // (apt_set_debug("grid_walk", 1): measurement knob, the nested item walk of render_frame_kernel only.)
        QueueArgs qa;
        uint64_t waves;
        size_t qlds;
        const bool rrk = ta.rr_start != 0;
        if (!queue_launch_shape(dbg, lp, rrk, true, pixel_count, retire, qa, waves, qlds)) return fail(APT_ERR_ARG, "pixel_count too large for one launch; shard it%s");
        // (the walk statistics behind apt_set_trace_counter are a template flag: a frame without a counter does not carry them)
        const bool oracle = p->mode == APT_MODE_ORACLE, stats = ta.traced != nullptr;
        const dim3 qgrid((unsigned)waves), qblock(64);
        // APT_FLAG_GRID_SLOTS: the caller vouches for the grid (apt_grid_flags): this launch is the frame's only one, and a grid that does not
        // keep the promise is reported through the status word (grid_walk == 3 tells the kernel to report instead of returning silently)
        // (Only with a status word to report through: without one -- a context's first launch inside a stream capture, a device index beyond
        // the context's table -- a grid that breaks the promise would render nothing and say nothing, so the two-launch form is kept then.)
        const bool vouched = (p->flags & APT_FLAG_GRID_SLOTS) && eps_allows_rootkey(p->eps) && ta.status != nullptr;
        TraceArgs ta_q = ta;
        ta_q.grid_walk = vouched ? 3u : 0u;
#define APT_LAUNCH_GRID_QUEUE(M, R, S) hipLaunchKernelGGL((render_frame_queue8_kernel<M, R, kSceneGrid, S>), qgrid, qblock, qlds, st, spheres, fa, ta_q, lp, qa)
        if (oracle) {
            if (rrk) { if (stats) APT_LAUNCH_GRID_QUEUE(kModeOracle, true, true); else APT_LAUNCH_GRID_QUEUE(kModeOracle, true, false); }
            else { if (stats) APT_LAUNCH_GRID_QUEUE(kModeOracle, false, true); else APT_LAUNCH_GRID_QUEUE(kModeOracle, false, false); }
        } else {
            if (rrk) { if (stats) APT_LAUNCH_GRID_QUEUE(kModeKernel, true, true); else APT_LAUNCH_GRID_QUEUE(kModeKernel, true, false); }
            else { if (stats) APT_LAUNCH_GRID_QUEUE(kModeKernel, false, true); else APT_LAUNCH_GRID_QUEUE(kModeKernel, false, false); }
        }
#undef APT_LAUNCH_GRID_QUEUE
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e);
        if (vouched) return APT_OK;
        ta_frame.grid_walk = 2;
    }
    size_t lds = lp.nleaves > 1 ? (size_t)kMaxStack * 3 * kStackSlots * sizeof(float) : 0;
    if (retire && !ns8 && !ta.grid && group == 8) lds += (size_t)(kBlock / 64) * 3 * 8 * lp.maxleaf * sizeof(float); // colour queue of the LDS-tile form
    const dim3 grid((unsigned)blocks);
    const int sck = ns8 ? kScene8 : (ta.grid ? kSceneGrid : kSceneTiles);
    if (p->mode == APT_MODE_ORACLE) {
        if (sck == kScene8) launch_frame_g<kModeOracle, kScene8>(group, retire, grid, lds, st, spheres, fa, ta, lp);
        else if (sck == kSceneGrid) launch_frame_g<kModeOracle, kSceneGrid>(group, retire, grid, lds, st, spheres, fa, ta_frame, lp);
        else launch_frame_g<kModeOracle, kSceneTiles>(group, retire, grid, lds, st, spheres, fa, ta, lp);
    } else {
        if (sck == kScene8) launch_frame_g<kModeKernel, kScene8>(group, retire, grid, lds, st, spheres, fa, ta, lp);
        else if (sck == kSceneGrid) launch_frame_g<kModeKernel, kSceneGrid>(group, retire, grid, lds, st, spheres, fa, ta_frame, lp);
        else launch_frame_g<kModeKernel, kSceneTiles>(group, retire, grid, lds, st, spheres, fa, ta, lp);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? APT_OK : hip_fail(e);
}

// contiguous near-equal split of [0,total) into `parts`: part r -> (begin, count); the first total%parts get one more
void split_range(uint64_t total, uint64_t r, uint64_t parts, uint64_t &begin, uint64_t &count) {
    const uint64_t base = total / parts, extra = total % parts;
    begin = r * base + (r < extra ? r : extra);
    count = base + (r < extra ? 1 : 0);
}

} // namespace

// ---- one process, several GPUs (include/render_mi355x.h: apt_multi_*) ---------------------------
// The frame is cut into `bands * stripes` contiguous stripes of x-major pixels; band b renders stripes
// b, b+bands, b+2*bands, ... on its own device and stream, and every stripe is copied straight into the full
// framebuffer on the root device with hipMemcpyPeerAsync (xGMI: each peer has its own link to the root, so the
// copies of different bands run in parallel; no ring, no collective needed for a gather to one root).
struct apt_multi {
    struct Band {
        int device = 0;
        hipStream_t stream = nullptr;
        hipEvent_t start = nullptr, stop = nullptr;
        float *sph = nullptr;          // the scene table on this device
        float *fb = nullptr;           // [3][max stripe] per stripe, stripes back to back
        uint8_t *u8 = nullptr;
        uint64_t pixels = 0;           // total pixels of this band
    };
    std::vector<Band> bands;
    apt_render_params params;
    uint32_t stripes = 1;
    uint64_t max_stripe = 0;
    int root = 0;
};

extern "C" {

int apt_abi_version(void) { return APT_ABI_VERSION; }

int apt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

// ---- contexts ---------------------------------------------------------------------------------------
apt_context *apt_context_create(void) { clear_error(); return new (std::nothrow) apt_context(); }
void apt_context_destroy(apt_context *ctx) {
    clear_error();
    if (!ctx) return;
    uint32_t *words[apt::kMaxStatusDevices];
    ctx->status_release(words);
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (int d = 0; d < apt::kMaxStatusDevices; ++d)
        if (words[d] && hipSetDevice(d) == hipSuccess) (void)hipFree(words[d]);
    if (have_cur) (void)hipSetDevice(cur);
    (void)hipGetLastError();
    delete ctx;
}

int apt_context_set_debug(apt_context *ctx, const char *key, double value) {
    clear_error();
    if (!ctx) return fail(APT_ERR_ARG, "context is null%s");
    return ctx->set_debug(key, value);
}

int apt_context_get_debug(apt_context *ctx, const char *key, double *value) {
    clear_error();
    if (!ctx || !value) return fail(APT_ERR_ARG, "context/value is null%s");
    return ctx->get_debug(key, value);
}

// Reads and clears the status word of the current device (see the header).  Synchronises `stream` first.
int apt_context_check(apt_context *ctx, void *stream) {
    clear_error();
    if (!ctx) return fail(APT_ERR_ARG, "context is null%s");
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e);
    uint32_t *w = status_word(*ctx, (hipStream_t)stream);
    if (!w) return APT_OK;                       // no word could be made: nothing was ever reported into it
    uint32_t bits = 0;
    e = hipMemcpy(&bits, w, sizeof bits, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return hip_fail(e);
    if (!bits) return APT_OK;
    e = hipMemset(w, 0, sizeof bits);
    if (e != hipSuccess) (void)hipGetLastError();
    std::string what;
    if (bits & APT_DEV_QUEUE_GUARD) what += " queue-loop-bound";
    if (bits & APT_DEV_GRID_TURNS) what += " grid-walk-bound";
    if (bits & APT_DEV_LDS_BASE) what += " lds-base";
    if (bits & APT_DEV_GRID_MISMATCH) what += " grid-mismatch";
    if (bits & ~(uint32_t)(APT_DEV_QUEUE_GUARD | APT_DEV_GRID_TURNS | APT_DEV_LDS_BASE | APT_DEV_GRID_MISMATCH)) what += " unknown-bits";
    return fail(APT_ERR_DEVICE, "a kernel reported a failure through the device status word:%s (the frame it wrote is incomplete)", what.c_str());
}

int apt_context_set_params(apt_context *ctx, const apt_render_params *p) {
    clear_error();
    if (!ctx) return fail(APT_ERR_ARG, "context is null%s");
    int rc = check_params(p);
    if (rc) return rc;
    ctx->set_params(*p);
    return APT_OK;
}

int apt_context_set_trace_counter(apt_context *ctx, uint64_t *device_counter) {
    clear_error();
    if (!ctx) return fail(APT_ERR_ARG, "context is null%s");
    ctx->set_trace_counter((unsigned long long *)device_counter);
    return APT_OK;
}

int apt_context_set_refill_lanes(apt_context *ctx, uint32_t lanes) {
    clear_error();
    if (!ctx) return fail(APT_ERR_ARG, "context is null%s");
    if (lanes < 1 || lanes > 64) return fail(APT_ERR_ARG, "refill lanes: 1..64%s");
    ctx->set_refill_lanes(lanes);
    return APT_OK;
}

void apt_context_render_do(apt_context *ctx, uint32_t blockDim, void *l2ctrl, void *stream, uint8_t *rays,
                           uint8_t *spheres, uint8_t *colors) {
    clear_error();
    (void)blockDim; // the reference's 8-way partition (render.cpp:9-10,24): results do not depend on it
    (void)l2ctrl;
    if (!ctx) { (void)fail(APT_ERR_ARG, "context is null%s"); return; }
    const Launch ls = launch_state(*ctx, stream);
    (void)do_render_paths(ls, &ls.cv.params, stream, (const float *)rays, (const float *)spheres, (float *)colors);
}

int apt_context_render_do_ex(apt_context *ctx, const apt_render_params *p, void *stream, const float *rays,
                             const float *spheres, float *colors) {
    clear_error();
    if (!ctx) return fail(APT_ERR_ARG, "context is null%s");
    return do_render_paths(launch_state(*ctx, stream), p, stream, rays, spheres, colors);
}

int apt_context_render_frame(apt_context *ctx, const apt_render_params *p, void *stream, const float *spheres,
                             uint64_t pixel_begin, uint64_t pixel_count, float *fb, uint8_t *fb_u8) {
    clear_error();
    if (!ctx) return fail(APT_ERR_ARG, "context is null%s");
    return do_render_frame(launch_state(*ctx, stream), p, stream, spheres, pixel_begin, pixel_count, fb, fb_u8);
}

// ---- the context-free forms: the process-wide default context -----------------------------------------
int apt_set_default_params(const apt_render_params *p) { return apt_context_set_params(&apt::default_context(), p); }
int apt_set_refill_lanes(uint32_t lanes) { return apt_context_set_refill_lanes(&apt::default_context(), lanes); }
int apt_set_trace_counter(uint64_t *device_counter) { return apt_context_set_trace_counter(&apt::default_context(), device_counter); }
int apt_set_debug(const char *key, double value) { return apt_context_set_debug(&apt::default_context(), key, value); }
int apt_get_debug(const char *key, double *value) { return apt_context_get_debug(&apt::default_context(), key, value); }
int apt_check(void *stream) { return apt_context_check(&apt::default_context(), stream); }

int render_do_ex(const apt_render_params *p, void *stream, const float *rays, const float *spheres, float *colors) {
    return apt_context_render_do_ex(&apt::default_context(), p, stream, rays, spheres, colors);
}

void render_do(uint32_t blockDim, void *l2ctrl, void *stream, uint8_t *rays, uint8_t *spheres, uint8_t *colors) {
    apt_context_render_do(&apt::default_context(), blockDim, l2ctrl, stream, rays, spheres, colors);
}

// The same function under a name a C++ translation unit can bind next to its own C++-linkage render_do
// (render_do_cxx.cpp: the reference declares render_do WITHOUT extern "C", src/main.cpp:9-10).
void apt_render_do(uint32_t blockDim, void *l2ctrl, void *stream, uint8_t *rays, uint8_t *spheres, uint8_t *colors) {
    apt_context_render_do(&apt::default_context(), blockDim, l2ctrl, stream, rays, spheres, colors);
}

int render_frame(const apt_render_params *p, void *stream, const float *spheres, uint64_t pixel_begin,
                 uint64_t pixel_count, float *fb, uint8_t *fb_u8) {
    return apt_context_render_frame(&apt::default_context(), p, stream, spheres, pixel_begin, pixel_count, fb, fb_u8);
}

// The CPU-simulator shape of the boundary (src/main.cpp:21-44: ICPU_RUN_KF(render, blockDim, rays, spheres,
// colors) on HOST buffers, synchronous): copies in, renders with the default context's parameters, copies out.
int apt_render_host(uint32_t blockDim, const uint8_t *rays, const uint8_t *spheres, uint8_t *colors) {
    clear_error();
    if (!rays || !spheres || !colors) return fail(APT_ERR_ARG, "rays/spheres/colors must be non-null%s");
    const Launch ls = launch_state(apt::default_context(), nullptr);
    const apt_render_params &p = ls.cv.params;
    const size_t n = (size_t)p.width * p.height * 4u * p.samples;
    // The reference's call renders the whole ray array (src/main.cpp:18-25); with a STRICT path sub-range in the default parameters the
    // colours outside it would be copied back from memory no kernel wrote.  (path_begin 0 with path_count 0 or N is the whole frame.)
    if (p.path_begin != 0 || (p.path_count != 0 && p.path_count != n)) return fail(APT_ERR_ARG, "apt_render_host: the default parameters carry a path sub-range; it renders whole frames only%s");
    const size_t sph_bytes = ((size_t)p.num_spheres * 10 + 127) / 128 * 128 * sizeof(float);
    float *d_rays = nullptr, *d_sph = nullptr, *d_col = nullptr;
    hipError_t e = hipMalloc(&d_rays, n * 24);
    if (e == hipSuccess) e = hipMalloc(&d_sph, sph_bytes);
    if (e == hipSuccess) e = hipMalloc(&d_col, n * 12);
    if (e == hipSuccess) e = hipMemcpy(d_rays, rays, n * 24, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_sph, spheres, sph_bytes, hipMemcpyHostToDevice);
    int rc = APT_OK;
    if (e == hipSuccess) {
        (void)blockDim;
        rc = do_render_paths(ls, &p, nullptr, d_rays, d_sph, d_col);
        if (rc == APT_OK) e = hipMemcpy(colors, d_col, n * 12, hipMemcpyDeviceToHost); // synchronises the null stream
    }
    (void)hipFree(d_rays); (void)hipFree(d_sph); (void)hipFree(d_col);
    if (rc != APT_OK) return rc;
    if (e != hipSuccess) return hip_fail(e);
    return apt_context_check(&apt::default_context(), nullptr);   // a kernel may have reported a failure (clears and sets the error record itself)
}

// ---- one process, several GPUs ---------------------------------------------------------------------------
void apt_multi_destroy(apt_multi *m) {
    clear_error();
    if (!m) return;
    for (auto &b : m->bands) {
        if (hipSetDevice(b.device) != hipSuccess) continue;
        if (b.start) (void)hipEventDestroy(b.start);
        if (b.stop) (void)hipEventDestroy(b.stop);
        if (b.stream) (void)hipStreamDestroy(b.stream);
        (void)hipFree(b.sph); (void)hipFree(b.fb); (void)hipFree(b.u8);
    }
    (void)hipSetDevice(m->root);
    delete m;
}

int apt_multi_create(const int *device_ids, uint32_t num_bands, uint32_t stripes, const apt_render_params *p,
                     const float *spheres_host, apt_multi **out) {
    clear_error();
    if (!device_ids || !num_bands || !stripes || !spheres_host || !out) return fail(APT_ERR_ARG, "apt_multi_create: bad arguments%s");
    int rc = check_params(p);
    if (rc) return rc;
    if (p->accel) return fail(APT_ERR_ARG, "apt_multi_create: accel is a single-device address; not supported here%s");
    const int ndev = apt_device_count();
    for (uint32_t b = 0; b < num_bands; ++b)
        if (device_ids[b] < 0 || device_ids[b] >= ndev) return fail(APT_ERR_DEVICE, "apt_multi_create: device id out of range%s");
    const uint64_t npix = (uint64_t)p->width * p->height, parts = (uint64_t)num_bands * stripes;
    if (parts > npix) return fail(APT_ERR_ARG, "apt_multi_create: more stripes than pixels%s");
    apt_multi *m = new (std::nothrow) apt_multi();
    if (!m) return fail(APT_ERR_DEVICE, "out of host memory%s");
    m->params = *p; m->stripes = stripes; m->root = device_ids[0];
    uint64_t b0, c0;
    split_range(npix, 0, parts, b0, c0);
    m->max_stripe = c0;
    m->bands.resize(num_bands);
    const size_t sph_bytes = ((size_t)p->num_spheres * 10 + 127) / 128 * 128 * sizeof(float);
    hipError_t e = hipSuccess;
    for (uint32_t b = 0; b < num_bands && e == hipSuccess; ++b) {
        apt_multi::Band &bd = m->bands[b];
        bd.device = device_ids[b];
        e = hipSetDevice(bd.device);
        if (e == hipSuccess && bd.device != m->root) { // let the root's copy engine read this device (no-op if already on)
            int can = 0;
            (void)hipDeviceCanAccessPeer(&can, bd.device, m->root);
            if (can) { hipError_t pe = hipDeviceEnablePeerAccess(m->root, 0); if (pe != hipSuccess) (void)hipGetLastError(); }
        }
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&bd.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreate(&bd.start);
        if (e == hipSuccess) e = hipEventCreate(&bd.stop);
        if (e == hipSuccess) e = hipMalloc(&bd.sph, sph_bytes);
        if (e == hipSuccess) e = hipMemcpy(bd.sph, spheres_host, sph_bytes, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc(&bd.fb, (size_t)stripes * 3 * m->max_stripe * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(&bd.u8, (size_t)stripes * 3 * m->max_stripe);
    }
    (void)hipSetDevice(m->root);
    if (e != hipSuccess) { rc = hip_fail(e); apt_multi_destroy(m); apt::set_error(rc, "apt_multi_create: HIP: %s", hipGetErrorString(e)); return rc; }
    *out = m;
    return APT_OK;
}

int apt_multi_render(apt_multi *m, float *fb_root, uint8_t *u8_root, float *band_kernel_ms) {
    clear_error();
    if (!m || !fb_root) return fail(APT_ERR_ARG, "apt_multi_render: handle/fb must be non-null%s");
    const uint64_t npix = (uint64_t)m->params.width * m->params.height;
    const uint64_t nb = m->bands.size(), parts = nb * m->stripes;
    int rc = APT_OK;
    hipError_t e = hipSuccess;
    for (uint64_t b = 0; b < nb && rc == APT_OK && e == hipSuccess; ++b) {
        apt_multi::Band &bd = m->bands[b];
        e = hipSetDevice(bd.device);
        if (e != hipSuccess) break;
        Launch ls = launch_state(apt::default_context(), bd.stream);   // (the status word is per device: taken with the band's device current)
        // the statistics block of the default context is an address on ITS device: only bands on the root device may count into it
        if (bd.device != m->root) ls.cv.trace_counter = nullptr;
        e = hipEventRecord(bd.start, bd.stream);
        for (uint32_t s = 0; s < m->stripes && rc == APT_OK && e == hipSuccess; ++s) {
            uint64_t begin, count;
            split_range(npix, (uint64_t)s * nb + b, parts, begin, count);   // interleaved: stripe s*nb + b belongs to band b
            float *fb = bd.fb + (size_t)s * 3 * m->max_stripe;
            uint8_t *u8 = bd.u8 + (size_t)s * 3 * m->max_stripe;
            rc = do_render_frame(ls, &m->params, bd.stream, bd.sph, begin, count, fb, u8_root ? u8 : nullptr);
            if (rc != APT_OK) break;
            if (s + 1 == m->stripes) e = hipEventRecord(bd.stop, bd.stream);  // kernels only: the copies follow
        }
        for (uint32_t s = 0; s < m->stripes && rc == APT_OK && e == hipSuccess; ++s) {
            uint64_t begin, count;
            split_range(npix, (uint64_t)s * nb + b, parts, begin, count);
            const float *fb = bd.fb + (size_t)s * 3 * m->max_stripe;
            const uint8_t *u8 = bd.u8 + (size_t)s * 3 * m->max_stripe;
            for (int ch = 0; ch < 3 && e == hipSuccess; ++ch)   // band-local planes [3][count] -> planes of the full frame
                e = hipMemcpyPeerAsync(fb_root + (size_t)ch * npix + begin, m->root, fb + (size_t)ch * count, bd.device,
                                       count * sizeof(float), bd.stream);
            if (u8_root && e == hipSuccess)
                e = hipMemcpyPeerAsync(u8_root + begin * 3, m->root, u8, bd.device, count * 3, bd.stream);
        }
    }
    for (auto &bd : m->bands) { // wait for every band (also after an error: nothing may still be writing)
        if (hipSetDevice(bd.device) == hipSuccess) { hipError_t se = hipStreamSynchronize(bd.stream); if (e == hipSuccess) e = se; }
    }
    if (band_kernel_ms && rc == APT_OK && e == hipSuccess)
        for (uint64_t b = 0; b < nb; ++b) {
            (void)hipSetDevice(m->bands[b].device);
            if (hipEventElapsedTime(&band_kernel_ms[b], m->bands[b].start, m->bands[b].stop) != hipSuccess) band_kernel_ms[b] = -1.0f;
        }
    int dev_rc = APT_OK;
    std::string dev_msg;
    if (rc == APT_OK && e == hipSuccess)   // the device status words of every device that rendered (each check clears its word)
        for (auto &bd : m->bands)
            if (hipSetDevice(bd.device) == hipSuccess && apt_context_check(&apt::default_context(), bd.stream) != APT_OK && dev_rc == APT_OK) {
                dev_rc = apt_last_status();
                dev_msg = apt_last_error();
            }
    (void)hipSetDevice(m->root);
    if (rc != APT_OK) return rc;             // (the error record still describes the launch that failed: no check ran after it)
    if (e != hipSuccess) return hip_fail(e);
    return dev_rc == APT_OK ? APT_OK : fail(dev_rc, "apt_multi_render: %s", dev_msg.c_str());
}

int apt_build_grid_device(const float *spheres_dev, uint32_t ns, void *stream, void *grid_dev, size_t capacity,
                          size_t *out_bytes) {
    clear_error();
    if (!spheres_dev || ns == 0 || !out_bytes) return fail(APT_ERR_ARG, "apt_build_grid_device: spheres/out_bytes must be non-null, num_spheres non-zero%s");
    hipStream_t st = (hipStream_t)stream;
    // workspace: radii, the two ordered lists, statistics (freed on return; a build step, not a render call)
    float *rad = nullptr;
    uint32_t *large = nullptr, *small = nullptr, *count = nullptr, *cursor = nullptr, *sums = nullptr, *long_count = nullptr;
    uint2 *long_cells = nullptr;
    bool break_out = false;
    GridBuildStats *stats_d = nullptr;
    hipError_t e = hipMalloc(&rad, (size_t)ns * 4);
    if (e == hipSuccess) e = hipMalloc(&large, (size_t)ns * 4);
    if (e == hipSuccess) e = hipMalloc(&small, (size_t)ns * 4);
    if (e == hipSuccess) e = hipMalloc(&stats_d, sizeof(GridBuildStats));
    GridBuildStats stt;
    int rc = APT_OK;
    GridHeader h;
    size_t words = 0;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(grid_classify_kernel, dim3(1), dim3(kGB), 0, st, spheres_dev, ns, rad, large, small, stats_d);
        e = hipMemcpyAsync(&stt, stats_d, sizeof stt, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (e == hipSuccess) {
        const double knob = apt::default_context().snapshot().debug.grid_spheres_per_cell;   // apt_set_debug("grid_spheres_per_cell", v)
        const double per_cell = knob > 0.0 ? knob : kGridSpheresPerCell;
        grid_header_from_stats(ns, stt.nsmall, stt.nlarge, stt.lo, stt.hi, stt.scale, per_cell, h);
        const uint64_t nc1 = (uint64_t)h.ncells + 1, nblk = (nc1 + kGB - 1) / kGB;
        e = hipMalloc(&count, nc1 * 4);
        if (e == hipSuccess) e = hipMalloc(&cursor, (size_t)h.ncells * 4);
        if (e == hipSuccess) e = hipMalloc(&sums, nblk * 4);
        if (e == hipSuccess) e = hipMemsetAsync(count, 0, nc1 * 4, st);
        if (e == hipSuccess) e = hipMemsetAsync(cursor, 0, (size_t)h.ncells * 4, st);
        uint32_t nitems = 0;
        if (e == hipSuccess) {
            if (stt.nsmall) hipLaunchKernelGGL(grid_count_kernel, dim3((stt.nsmall + 255) / 256), dim3(256), 0, st, h, spheres_dev, rad, small, stt.nsmall, count);
            hipLaunchKernelGGL(scan_blocks_kernel, dim3((unsigned)nblk), dim3(kGB), 0, st, count, nc1, sums);
            hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(kGB), 0, st, sums, (uint32_t)nblk);
            hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)nblk), dim3(kGB), 0, st, count, nc1, sums);
            e = hipMemcpyAsync(&nitems, count + h.ncells, 4, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
        }
        if (e == hipSuccess) {
            words = grid_header_offsets(h, nitems);
            *out_bytes = words * 4;
            if (!grid_dev) rc = APT_OK;                                       // size query
            else if (capacity < words * 4) rc = fail(APT_ERR_ARG, "apt_build_grid_device: capacity too small (see *out_bytes)%s");
            else {
                uint32_t *w = (uint32_t *)grid_dev;
                e = hipMemsetAsync(w, 0, words * 4, st);
                if (e == hipSuccess) e = hipMemcpyAsync(w, &h, sizeof h, hipMemcpyHostToDevice, st);
                if (e == hipSuccess && h.nlarge) e = hipMemcpyAsync(w + h.off_large, large, (size_t)h.nlarge * 4, hipMemcpyDeviceToDevice, st);
                if (e == hipSuccess) e = hipMemcpyAsync(w + h.off_cells, count, nc1 * 4, hipMemcpyDeviceToDevice, st);
                if (e == hipSuccess) {
                    if (stt.nsmall) hipLaunchKernelGGL(grid_fill_kernel, dim3((stt.nsmall + 255) / 256), dim3(256), 0, st, h, spheres_dev, rad, small, stt.nsmall, count, cursor, w + h.off_items);
                    // short cell lists are sorted on the device, long ones (clustered scenes) reported and sorted here
                    const uint32_t max_long = nitems / (kGridSortInline + 1u) + 1u;
                    e = hipMalloc(&long_cells, (size_t)max_long * sizeof(uint2));
                    if (e == hipSuccess) e = hipMalloc(&long_count, 4);
                    if (e == hipSuccess) e = hipMemsetAsync(long_count, 0, 4, st);
                    uint32_t nlong = 0;
                    if (e == hipSuccess) {
                        hipLaunchKernelGGL(grid_sort_cells_kernel, dim3((h.ncells + 255) / 256), dim3(256), 0, st, h.ncells, count, w + h.off_items, long_count, long_cells);
                        e = hipMemcpyAsync(&nlong, long_count, 4, hipMemcpyDeviceToHost, st);
                        if (e == hipSuccess) e = hipStreamSynchronize(st);
                    }
                    if (e == hipSuccess && nlong) {
                        std::vector<uint2> cells(nlong);
                        std::vector<uint32_t> host_items(nitems);
                        e = hipMemcpy(cells.data(), long_cells, (size_t)nlong * sizeof(uint2), hipMemcpyDeviceToHost);
                        if (e == hipSuccess) e = hipMemcpy(host_items.data(), w + h.off_items, (size_t)nitems * 4, hipMemcpyDeviceToHost);
                        if (e == hipSuccess) {
                            for (const uint2 &c : cells) std::sort(host_items.begin() + c.x, host_items.begin() + c.y);
                            // only the long cells go back: the short ones were sorted in place by the kernel above (the copy out saw them sorted already)
                            e = hipMemcpy(w + h.off_items, host_items.data(), (size_t)nitems * 4, hipMemcpyHostToDevice);
                        }
                    }
                    if (e != hipSuccess) break_out = true;
                    const uint32_t ng = ns > nitems ? ns : nitems;
                    if (!break_out) hipLaunchKernelGGL(grid_geom_kernel, dim3((ng + 255) / 256), dim3(256), 0, st, spheres_dev, ns, w + h.off_items, nitems,
                                       reinterpret_cast<float4 *>(w + h.off_geom), reinterpret_cast<float4 *>(w + h.off_item_geom));
                    if (!break_out && h.off_cellslot) hipLaunchKernelGGL(grid_slots_kernel, dim3((std::max(apt::grid_bordered_cells(h.n) + 1u, ns) + 255u) / 256u), dim3(256), 0, st, w, h, spheres_dev);
                    if (e == hipSuccess) e = hipGetLastError();
                    if (e == hipSuccess) e = hipStreamSynchronize(st);       // the workspace is freed below
                }
            }
        }
    }
    (void)hipFree(rad); (void)hipFree(large); (void)hipFree(small); (void)hipFree(stats_d);
    (void)hipFree(count); (void)hipFree(cursor); (void)hipFree(sums); (void)hipFree(long_count); (void)hipFree(long_cells);
    if (rc != APT_OK) return rc;
    return e == hipSuccess ? APT_OK : hip_fail(e);
}

int apt_selftest_sqrt(int variant, void *stream, uint64_t first_bits, uint64_t count, uint64_t *device_result2) {
    clear_error();
    if (!device_result2 || variant < 0 || variant > 3) return fail(APT_ERR_ARG, "apt_selftest_sqrt: bad arguments%s");
    if (first_bits + count > (1ull << 32)) return fail(APT_ERR_ARG, "apt_selftest_sqrt: range beyond 2^32%s");
    if (count == 0) return APT_OK;
    hipLaunchKernelGGL(selftest_sqrt_kernel, dim3(256 * 16), dim3(kBlock), 0, (hipStream_t)stream, variant, first_bits,
                       count, (unsigned long long *)device_result2);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? APT_OK : hip_fail(e);
}

int apt_selftest_div3(void *stream, uint64_t first, uint64_t count, uint64_t *device_result3) {
    clear_error();
    if (!device_result3) return fail(APT_ERR_ARG, "apt_selftest_div3: bad arguments%s");
    if (count == 0) return APT_OK;
    hipLaunchKernelGGL(selftest_div3_kernel, dim3(256 * 16), dim3(kBlock), 0, (hipStream_t)stream, first, count,
                       (unsigned long long *)device_result3);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? APT_OK : hip_fail(e);
}

int apt_test_scene(const apt_render_params *p, void *stream, const float *rays, const float *spheres, float *out) {
    clear_error();
    int rc = check_params(p);
    if (rc) return rc;
    if (!rays || !spheres || !out) return fail(APT_ERR_ARG, "rays/spheres/out must be non-null%s");
    const uint64_t n = (uint64_t)p->width * p->height * 4u * p->samples;
    const uint64_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return fail(APT_ERR_ARG, "image too large for one launch%s");
    hipLaunchKernelGGL(test_scene_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, rays, spheres,
                       out, n, p->num_spheres, p->light_index, p->eps);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? APT_OK : hip_fail(e);
}

int apt_gen_rays_device(const apt_render_params *p, void *stream, float *rays) {
    clear_error();
    int rc = check_params(p);
    if (rc) return rc;
    if (!rays) return fail(APT_ERR_ARG, "rays must be non-null%s");
    const uint64_t n = (uint64_t)p->width * p->height * 4u * p->samples;
    const uint64_t b = p->path_begin;
    if (b > n) return fail(APT_ERR_ARG, "path_begin beyond the image%s");
    const uint64_t c = p->path_count ? p->path_count : n - b;
    if (b + c > n) return fail(APT_ERR_ARG, "path range beyond the image%s");
    if (c == 0) return APT_OK;
    const uint64_t blocks = (c + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return fail(APT_ERR_ARG, "path_count too large for one launch; shard it%s");
    Camera cam;
    camera_init(cam, p->width, p->height);
    const bool band = p->flags & APT_FLAG_BAND_BUFFERS;
    hipLaunchKernelGGL(gen_rays_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, cam, p->width,
                       p->height, p->samples, p->seed, band ? c : n, b, c, band ? rays - b : rays);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? APT_OK : hip_fail(e);
}

int apt_gen_rays_mt_device_ex(const apt_render_params *p, void *stream, const uint32_t *checkpoints, uint32_t stride,
                              uint64_t num_checkpoints, uint64_t first_block, float *rays) {
    clear_error();
    int rc = check_params(p);
    if (rc) return rc;
    if (!rays || !checkpoints || stride == 0) return fail(APT_ERR_ARG, "rays/checkpoints must be non-null, stride > 0%s");
    const uint64_t n = (uint64_t)p->width * p->height * 4u * p->samples;
    const uint64_t b = p->path_begin;
    if (b > n) return fail(APT_ERR_ARG, "path_begin beyond the image%s");
    const uint64_t c = p->path_count ? p->path_count : n - b;
    if (b + c > n) return fail(APT_ERR_ARG, "path range beyond the image%s");
    if (c == 0) return APT_OK;
    const uint64_t num_blocks = (n + kPathsPerBlock - 1) / kPathsPerBlock;           // of the whole stream
    const uint64_t blk_lo = b / kPathsPerBlock, blk_hi = (b + c + kPathsPerBlock - 1) / kPathsPerBlock; // blocks the range touches
    if (blk_lo < first_block) return fail(APT_ERR_ARG, "path range starts before the checkpoint window%s");
    const uint64_t cp_lo = (blk_lo - first_block) / stride, cp_hi = (blk_hi - first_block + stride - 1) / stride;
    if (cp_hi > num_checkpoints) return fail(APT_ERR_ARG, "not enough MT19937 checkpoints for this path range%s");
    if (cp_hi - cp_lo > 0x7fffffffull) return fail(APT_ERR_ARG, "too many checkpoints for one launch%s");
    Camera cam;
    camera_init(cam, p->width, p->height);
    const bool band = p->flags & APT_FLAG_BAND_BUFFERS;
    // one workgroup per checkpoint the range touches (workgroups of untouched checkpoints would only skip)
    hipLaunchKernelGGL(gen_rays_mt_kernel, dim3((unsigned)(cp_hi - cp_lo)), dim3(kBlock), 0, (hipStream_t)stream,
                       checkpoints + cp_lo * kMtN, stride, first_block + cp_lo * stride, num_blocks, cam, p->width, p->height,
                       p->samples, band ? c : n, b, b + c, band ? rays - b : rays);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? APT_OK : hip_fail(e);
}

int apt_gen_rays_mt_device(const apt_render_params *p, void *stream, const uint32_t *checkpoints, uint32_t stride,
                           uint64_t num_checkpoints, float *rays) {
    return apt_gen_rays_mt_device_ex(p, stream, checkpoints, stride, num_checkpoints, 0, rays);
}

int apt_render_frame_mt(const apt_render_params *p, void *stream, const uint32_t *checkpoints, uint64_t num_checkpoints,
                        uint64_t first_group, const float *spheres, uint64_t pixel_begin, uint64_t pixel_count, float *fb, uint8_t *fb_u8) {
    clear_error();
    int rc = check_params(p);
    if (rc) return rc;
    if (!checkpoints || !spheres || !fb) return fail(APT_ERR_ARG, "apt_render_frame_mt: checkpoints/spheres/fb must be non-null%s");
    if (p->num_spheres != 8) return fail(APT_ERR_SCENE, "apt_render_frame_mt: the 8-sphere scene only%s");
    if (p->flags & APT_FLAG_RR) return fail(APT_ERR_ARG, "apt_render_frame_mt: APT_FLAG_RR is not part of the reference's pipeline%s");
    uint32_t log2_s = 0;
    while ((1u << log2_s) < p->samples) ++log2_s;
    // 78 pixels are 2 * samples generator blocks for every sample count; samples in {8, 16, ..., 256} take the kernel whose sums are a
    // fixed 24 lanes per run, every other count (1, 2, 4 -- the reference's default is 1 --, non-powers of two, > 256) the general one
    const bool pow2_kernel = (1u << log2_s) == p->samples && p->samples >= 8 && p->samples <= 256;
    LeafProg lp;
    if ((rc = make_leaf_prog(p->samples, lp))) return rc;
    const uint64_t npix = (uint64_t)p->width * p->height;
    if (pixel_begin > npix || pixel_count > npix - pixel_begin) return fail(APT_ERR_ARG, "pixel range beyond the image%s");
    if (pixel_count == 0) return APT_OK;
    const uint64_t g_lo = pixel_begin / kMtGroupPixels, g_hi = (pixel_begin + pixel_count + kMtGroupPixels - 1) / kMtGroupPixels;
    if (g_lo < first_group || g_hi - first_group > num_checkpoints) return fail(APT_ERR_ARG, "apt_render_frame_mt: the checkpoint table does not cover the pixel range%s");
    if (g_hi - g_lo > 0x7fffffffull) return fail(APT_ERR_ARG, "pixel_count too large for one launch; shard it%s");
    const TraceArgs ta = make_trace_args(p, launch_state(apt::default_context(), stream));
    FrameArgs fa;
    camera_init(fa.cam, p->width, p->height);
    fa.width = p->width; fa.height = p->height; fa.samples = p->samples; fa.seed = p->seed;
    fa.pixel_begin = pixel_begin; fa.pixel_count = pixel_count; fa.fb = fb; fa.fb_u8 = fb_u8;
    MtFrameArgs ma;
    ma.checkpoints = checkpoints + (g_lo - first_group) * 624; ma.first_group = g_lo; ma.log2_s = log2_s;
    const dim3 grid((unsigned)(g_hi - g_lo));
    if (pow2_kernel) {
        if (p->mode == APT_MODE_ORACLE) hipLaunchKernelGGL((render_frame_mt_kernel<kModeOracle>), grid, dim3(kBlock), 0, (hipStream_t)stream, spheres, fa, ta, ma);
        else hipLaunchKernelGGL((render_frame_mt_kernel<kModeKernel>), grid, dim3(kBlock), 0, (hipStream_t)stream, spheres, fa, ta, ma);
    } else {
        if (p->mode == APT_MODE_ORACLE) hipLaunchKernelGGL((render_frame_mt_any_kernel<kModeOracle>), grid, dim3(kBlock), 0, (hipStream_t)stream, spheres, fa, ta, ma, lp);
        else hipLaunchKernelGGL((render_frame_mt_any_kernel<kModeKernel>), grid, dim3(kBlock), 0, (hipStream_t)stream, spheres, fa, ta, ma, lp);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? APT_OK : hip_fail(e);
}

int apt_decode_color_band(const apt_render_params *p, void *stream, const float *colors, uint64_t pixel_count, float *fb,
                          uint8_t *fb_u8) {
    clear_error();
    int rc = check_params(p);
    if (rc) return rc;
    if (!colors || !fb) return fail(APT_ERR_ARG, "colors/fb must be non-null%s");
    if (pixel_count == 0) return APT_OK;
    LeafProg lp;
    if ((rc = make_leaf_prog(p->samples, lp))) return rc;
    const uint64_t npix = pixel_count;                    // the band is decoded like an image of pixel_count pixels
    const bool wide = p->samples >= 8;                    // 8 lanes per sub-pixel row: coalesced loads
    // few samples: 2 lanes per row with float4 loads (decode_color_kernel4); every row then starts at a multiple of 16 bytes from `colors`
    const bool quad = wide && p->samples <= 8u * kDecode4Blocks && p->samples % 4 == 0 && ((uintptr_t)colors & 15u) == 0;
    const uint64_t lanes = npix * 3 * 4 * (quad ? 2 : (wide ? 8 : 1));
    const uint64_t blocks = (lanes + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return fail(APT_ERR_ARG, "band too large for one launch%s");
    // grid caps measured on the C2 buffer (profiles/microbench/decode_rates.hip): 8-lane form 4.9 / 5.4 / 6.1 / 5.5 TB/s at 256 x 64 / 256 / 1024 /
    // one round for S = 64 (S = 256: 5.1 / 5.5 / 5.6 / 5.7; below 64 samples the smaller grid wins); float4 form 4.3 / 6.1 / 4.6 TB/s at S = 8 / 16 / 32 with its cap of 256 x 256 (8-lane form: 1.2 / 2.3 / 4.1)
    if (quad)
        hipLaunchKernelGGL(decode_color_kernel4, dim3((unsigned)std::min<uint64_t>(blocks, 256u * 256u)), dim3(kBlock), 0, (hipStream_t)stream, colors,
                           p->samples, npix, lp, fb, fb_u8);
    else if (wide)
        hipLaunchKernelGGL(decode_color_kernel8, dim3((unsigned)std::min<uint64_t>(blocks, p->samples >= 64 ? 256u * 1024u : 256u * 256u)), dim3(kBlock), 0, (hipStream_t)stream, colors,
                           p->samples, npix, lp, fb, fb_u8);
    else
        hipLaunchKernelGGL(decode_color_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, colors,
                           p->samples, npix, lp, fb, fb_u8);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? APT_OK : hip_fail(e);
}

int apt_decode_color_device(const apt_render_params *p, void *stream, const float *colors, float *fb, uint8_t *fb_u8) {
    clear_error();
    if (!p) return fail(APT_ERR_ARG, "params is null%s");
    return apt_decode_color_band(p, stream, colors, (uint64_t)p->width * p->height, fb, fb_u8);
}

} // extern "C"

