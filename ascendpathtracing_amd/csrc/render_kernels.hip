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
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/render_mi355x.h"
#include "pt_core.h"

#include "pt_kernels.h"

namespace {

// ---- host side ----------------------------------------------------------------------------
thread_local std::string g_err;
apt_render_params g_default;
bool g_default_init = false;
unsigned long long *g_trace_counter = nullptr;
uint32_t g_refill_lanes = kRefillLanes;

int fail(int code, const char *fmt, const char *detail = "") {
    char buf[256];
    snprintf(buf, sizeof buf, fmt, detail);
    g_err = buf;
    return code;
}

void build_leaves(uint32_t n, std::vector<std::pair<uint32_t, uint32_t>> &out) {
    if (n <= 128) { out.push_back({n, 0u}); return; }
    uint32_t n2 = n / 2;
    n2 -= n2 % 8;
    build_leaves(n2, out);
    build_leaves(n - n2, out);
    out.back().second += 1;
}

int make_leaf_prog(uint32_t samples, LeafProg &lp) {
    std::vector<std::pair<uint32_t, uint32_t>> v;
    build_leaves(samples, v);
    if (v.size() > (size_t)kMaxLeaves) return fail(APT_ERR_ARG, "samples too large for the pairwise plan (max 8192)%s");
    memset(&lp, 0, sizeof lp);
    lp.nleaves = (uint32_t)v.size();
    for (size_t i = 0; i < v.size(); ++i) {
        lp.leaf[i] = v[i].first | (v[i].second << 16);
        lp.maxleaf = v[i].first > lp.maxleaf ? v[i].first : lp.maxleaf;
    }
    return APT_OK;
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

int hip_fail(hipError_t e) { return fail(APT_ERR_DEVICE, "HIP: %s", hipGetErrorString(e)); }

TraceArgs make_trace_args(const apt_render_params *p) {
    TraceArgs ta;
    ta.ns = p->num_spheres; ta.depth = p->depth; ta.light = p->light_index;
    ta.eps = p->eps; ta.gain = p->gain; ta.traced = g_trace_counter;
    ta.refill_lanes = g_refill_lanes;
    ta.grid = (p->num_spheres != 8) ? reinterpret_cast<const uint32_t *>((uintptr_t)p->accel) : nullptr;
    ta.emission = (p->flags & APT_FLAG_EMISSION) ? 1u : 0u;
    ta.rr_start = (p->flags & APT_FLAG_RR) ? (p->rr_start ? p->rr_start : 3u) : 0u;
    ta.seed = p->seed;
    return ta;
}

template <int MODE, int SC>
void launch_paths(bool retire, dim3 grid, hipStream_t st, const float *rays, const float *sph, float *colors,
                  uint64_t n, uint64_t b, uint64_t c, const TraceArgs &ta) {
    if (retire) hipLaunchKernelGGL((render_paths_kernel<MODE, SC, true>), grid, dim3(kBlock), 0, st, rays, sph, colors, n, b, c, ta);
    else hipLaunchKernelGGL((render_paths_kernel<MODE, SC, false>), grid, dim3(kBlock), 0, st, rays, sph, colors, n, b, c, ta);
}

template <int MODE, int SC, int GROUP>
void launch_frame(bool retire, dim3 grid, size_t lds, hipStream_t st, const float *sph, const FrameArgs &fa,
                  const TraceArgs &ta, const LeafProg &lp) {
    if (retire) hipLaunchKernelGGL((render_frame_kernel<MODE, SC, GROUP, true>), grid, dim3(kBlock), lds, st, sph, fa, ta, lp);
    else hipLaunchKernelGGL((render_frame_kernel<MODE, SC, GROUP, false>), grid, dim3(kBlock), lds, st, sph, fa, ta, lp);
}

template <int MODE, int SC>
void launch_frame_g(int group, bool retire, dim3 grid, size_t lds, hipStream_t st, const float *sph,
                    const FrameArgs &fa, const TraceArgs &ta, const LeafProg &lp) {
    if (group == 8) launch_frame<MODE, SC, 8>(retire, grid, lds, st, sph, fa, ta, lp);
    else launch_frame<MODE, SC, 1>(retire, grid, lds, st, sph, fa, ta, lp);
}

} // namespace

// =============================== C-ABI ======================================================
extern "C" {

int apt_abi_version(void) { return APT_ABI_VERSION; }
const char *apt_last_error(void) { return g_err.c_str(); }

int apt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

void apt_default_params(apt_render_params *p) {
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->struct_size = sizeof *p;
    p->width = 16; p->height = 16; p->samples = 1; // common.h:4-6
    p->depth = 5;                                   // render.cpp:141
    p->num_spheres = 8; p->light_index = 7;         // common.h:10, rt_helper.h:776
    p->eps = 1e-4f; p->gain = 12.0f;                // common.h:9, render.cpp:194
    p->mode = APT_MODE_KERNEL;
}

int apt_set_default_params(const apt_render_params *p) {
    int rc = check_params(p);
    if (rc) return rc;
    g_default = *p;
    g_default_init = true;
    return APT_OK;
}

int apt_selftest_sqrt(int variant, void *stream, uint64_t first_bits, uint64_t count, uint64_t *device_result2) {
    if (!device_result2 || variant < 0 || variant > 3) return fail(APT_ERR_ARG, "apt_selftest_sqrt: bad arguments%s");
    if (first_bits + count > (1ull << 32)) return fail(APT_ERR_ARG, "apt_selftest_sqrt: range beyond 2^32%s");
    if (count == 0) return APT_OK;
    hipLaunchKernelGGL(selftest_sqrt_kernel, dim3(256 * 16), dim3(kBlock), 0, (hipStream_t)stream, variant, first_bits,
                       count, (unsigned long long *)device_result2);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? APT_OK : hip_fail(e);
}

int apt_selftest_div3(void *stream, uint64_t first, uint64_t count, uint64_t *device_result3) {
    if (!device_result3) return fail(APT_ERR_ARG, "apt_selftest_div3: bad arguments%s");
    if (count == 0) return APT_OK;
    hipLaunchKernelGGL(selftest_div3_kernel, dim3(256 * 16), dim3(kBlock), 0, (hipStream_t)stream, first, count,
                       (unsigned long long *)device_result3);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? APT_OK : hip_fail(e);
}

int apt_set_refill_lanes(uint32_t lanes) {
    if (lanes < 1 || lanes > 64) return fail(APT_ERR_ARG, "apt_set_refill_lanes: 1..64%s");
    g_refill_lanes = lanes;
    return APT_OK;
}

int apt_set_trace_counter(uint64_t *device_counter) {
    g_trace_counter = (unsigned long long *)device_counter;
    return APT_OK;
}

int render_do_ex(const apt_render_params *p, void *stream, const float *rays, const float *spheres, float *colors) {
    int rc = check_params(p);
    if (rc) return rc;
    if (!rays || !spheres || !colors) return fail(APT_ERR_ARG, "rays/spheres/colors must be non-null%s");
    const uint64_t n = (uint64_t)p->width * p->height * 4u * p->samples;
    const uint64_t b = p->path_begin;
    if (b > n) return fail(APT_ERR_ARG, "path_begin beyond the image%s");
    const uint64_t c = p->path_count ? p->path_count : n - b;
    if (b + c > n) return fail(APT_ERR_ARG, "path range beyond the image%s");
    if (c == 0) return APT_OK;
    const uint64_t blocks = (c + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return fail(APT_ERR_ARG, "path_count too large for one launch; shard it%s");
    hipStream_t st = (hipStream_t)stream;
    const bool ns8 = p->num_spheres == 8;
    const TraceArgs ta = make_trace_args(p);
    const bool retire = p->flags & APT_FLAG_RETIRE;
    const dim3 grid((unsigned)blocks);
    const int sck = ns8 ? kScene8 : (ta.grid ? kSceneGrid : kSceneTiles);
    if (retire && ns8) { // wave-level queue: one wave per kQueueChunk consecutive paths
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

void render_do(uint32_t blockDim, void *l2ctrl, void *stream, uint8_t *rays, uint8_t *spheres, uint8_t *colors) {
    (void)blockDim; // the reference's 8-way partition (render.cpp:9-10,24): results do not depend on it
    (void)l2ctrl;
    if (!g_default_init) { apt_default_params(&g_default); g_default_init = true; }
    (void)render_do_ex(&g_default, stream, (const float *)rays, (const float *)spheres, (float *)colors);
}

int render_frame(const apt_render_params *p, void *stream, const float *spheres, uint64_t pixel_begin,
                 uint64_t pixel_count, float *fb, uint8_t *fb_u8) {
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
    const TraceArgs ta = make_trace_args(p);
    FrameArgs fa;
    camera_init(fa.cam, p->width, p->height);
    fa.width = p->width; fa.height = p->height; fa.samples = p->samples; fa.seed = p->seed;
    fa.pixel_begin = pixel_begin; fa.pixel_count = pixel_count; fa.fb = fb; fa.fb_u8 = fb_u8;
    const bool retire = p->flags & APT_FLAG_RETIRE;
    size_t lds = lp.nleaves > 1 ? (size_t)kMaxStack * 3 * kStackSlots * sizeof(float) : 0;
    if (retire && ns8 && group == 8) lds += (size_t)(kBlock / 64) * 3 * 8 * lp.maxleaf * sizeof(float); // colour queue
    const dim3 grid((unsigned)blocks);
    const int sck = ns8 ? kScene8 : (ta.grid ? kSceneGrid : kSceneTiles);
    if (p->mode == APT_MODE_ORACLE) {
        if (sck == kScene8) launch_frame_g<kModeOracle, kScene8>(group, retire, grid, lds, st, spheres, fa, ta, lp);
        else if (sck == kSceneGrid) launch_frame_g<kModeOracle, kSceneGrid>(group, retire, grid, lds, st, spheres, fa, ta, lp);
        else launch_frame_g<kModeOracle, kSceneTiles>(group, retire, grid, lds, st, spheres, fa, ta, lp);
    } else {
        if (sck == kScene8) launch_frame_g<kModeKernel, kScene8>(group, retire, grid, lds, st, spheres, fa, ta, lp);
        else if (sck == kSceneGrid) launch_frame_g<kModeKernel, kSceneGrid>(group, retire, grid, lds, st, spheres, fa, ta, lp);
        else launch_frame_g<kModeKernel, kSceneTiles>(group, retire, grid, lds, st, spheres, fa, ta, lp);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? APT_OK : hip_fail(e);
}

int apt_test_scene(const apt_render_params *p, void *stream, const float *rays, const float *spheres, float *out) {
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
    hipLaunchKernelGGL(gen_rays_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, cam, p->width,
                       p->height, p->samples, p->seed, n, b, c, rays);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? APT_OK : hip_fail(e);
}

int apt_gen_rays_mt_device(const apt_render_params *p, void *stream, const uint32_t *checkpoints, uint32_t stride,
                           uint64_t num_checkpoints, float *rays) {
    int rc = check_params(p);
    if (rc) return rc;
    if (!rays || !checkpoints || stride == 0) return fail(APT_ERR_ARG, "rays/checkpoints must be non-null, stride > 0%s");
    const uint64_t n = (uint64_t)p->width * p->height * 4u * p->samples;
    const uint64_t b = p->path_begin;
    if (b > n) return fail(APT_ERR_ARG, "path_begin beyond the image%s");
    const uint64_t c = p->path_count ? p->path_count : n - b;
    if (b + c > n) return fail(APT_ERR_ARG, "path range beyond the image%s");
    if (c == 0) return APT_OK;
    const uint64_t num_blocks = (n + kPathsPerBlock - 1) / kPathsPerBlock;
    const uint64_t need = (num_blocks + stride - 1) / stride;
    if (num_checkpoints < need) return fail(APT_ERR_ARG, "not enough MT19937 checkpoints for this image%s");
    if (need > 0x7fffffffull) return fail(APT_ERR_ARG, "too many checkpoints for one launch%s");
    Camera cam;
    camera_init(cam, p->width, p->height);
    hipLaunchKernelGGL(gen_rays_mt_kernel, dim3((unsigned)need), dim3(kBlock), 0, (hipStream_t)stream, checkpoints,
                       stride, num_blocks, cam, p->width, p->height, p->samples, n, b, b + c, rays);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? APT_OK : hip_fail(e);
}

int apt_decode_color_device(const apt_render_params *p, void *stream, const float *colors, float *fb, uint8_t *fb_u8) {
    int rc = check_params(p);
    if (rc) return rc;
    if (!colors || !fb) return fail(APT_ERR_ARG, "colors/fb must be non-null%s");
    LeafProg lp;
    if ((rc = make_leaf_prog(p->samples, lp))) return rc;
    const uint64_t npix = (uint64_t)p->width * p->height;
    const bool wide = p->samples >= 8;                    // 8 lanes per sub-pixel row: coalesced loads
    const uint64_t lanes = npix * 3 * 4 * (wide ? 8 : 1);
    const uint64_t blocks = (lanes + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return fail(APT_ERR_ARG, "image too large for one launch%s");
    if (wide)
        hipLaunchKernelGGL(decode_color_kernel8, dim3((unsigned)std::min<uint64_t>(blocks, 256u * 64u)), dim3(kBlock), 0, (hipStream_t)stream, colors,
                           p->samples, npix, lp, fb, fb_u8);
    else
        hipLaunchKernelGGL(decode_color_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, colors,
                           p->samples, npix, lp, fb, fb_u8);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? APT_OK : hip_fail(e);
}

} // extern "C"

