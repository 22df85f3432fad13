// pt_queue.h -- the fused frame kernel with active-ray compaction: render_frame_queue8_kernel.  Included by pt_kernels.h.
//   SC == kScene8     the reference's 8-sphere scene with APT_FLAG_RETIRE, samples >= 8 (what the text below describes)
//   SC == kSceneGrid  any scene behind the uniform grid's pair-slot tables, with or without APT_FLAG_RETIRE: the same pixel
//                     stream, ray pool and colour ring around a per-lane grid walk (run_grid(), near the end of the kernel)
//
// Round 3 rewrite of the wave-level sample queue (round 2: one ray slot per lane in registers, ray-generate for the
// whole wave whenever >= 32 lanes had an empty slot, the queue drained at every pairwise leaf; the flag returned
// 55-60 % of the work it removed).  What a wave does now:
//
//   * it owns `ppw` consecutive pixels and works through their samples as ONE stream of items (a "unit" = one
//     pairwise leaf of one pixel = 4 sub-pixels x n <= 128 samples); nothing is drained between units, so the only
//     idle lanes are those of the wave's very last unit;
//   * ray-generate (float64, ~230 VALU instructions per ray) is decoupled from tracing through a per-wave LDS RAY
//     POOL (64 entries x 32 bytes, FIFO; 32 entries in the grid form): when it is empty the wave generates a pool's worth
//     of consecutive items, one per lane, whatever the lanes' paths are doing (and serves the lanes that the pool's last
//     entries could not serve);
//   * after every bounce the lanes whose path is finished (alive bit cleared, throughput zero, depth reached: wave
//     masks on the scalar unit) park their throughput in the unit's colour buffer in LDS and take the next pool
//     entries: ballot -> mbcnt rank -> one exec-masked block of ds_reads straight into the path-state registers;
//     3 VALU instructions per bounce to detect finished paths, 7 for the whole refill, none for parking;
//   * a unit whose items have all been parked (all of them handed out -- the pool is a FIFO, so that is a count -- and
//     no running lane's colour address inside the unit's buffer: evaluated only when the generator needs the buffer) is
//     summed exactly as numpy's pairwise np.mean does it -- lane (sub, j) adds samples j, 8+j, ... in order, 3-step
//     butterfly, the n % 8 tail in order, leaves combined through a stack -- by the lanes of the wave, then
//     decoded (data_visualization.py:36-57).  Colour buffers form a ring of `nbuf` units, so tracing unit u+1 overlaps
//     the stragglers of unit u.
//
// The arithmetic of a bounce is bounce_ns8_v2 (pt_trace.h), the accumulation order is numpy's: the frame is bit-identical
// to the full-trace kernel's and to the CPU restatement's, and the traced-segment count equals the oracle's.
// A wave whose loop bound trips (a logic error; never seen) or whose LDS does not start at address 0 says so through the context's
// device status word (include/render_mi355x.h APT_DEV_*; the reference asserts inside its kernel, src/render.cpp:68-73).
//
// How the hot code is written (round 4): these kernels run at the issue rate of the vector pipe, so what they cost is their instruction
// COUNT.  Which lanes do what is kept in WAVE MASKS on the scalar unit; a per-lane `bool` made from a mask (v_cndmask + v_cmp) or a mask
// made from a bool set inside a divergent region costs two vector instructions each way, so the hot paths branch on the masks themselves
// (wave-uniform), compute unmasked -- an instruction costs its issue slot whatever the execution mask -- and write per-lane state through
// selects on the masks or short exec-masked asm groups (park / refill, roulette(), the grid form's turn()).
#pragma once
#include <type_traits>

#include "pt_trace.h"

namespace {

struct FrameArgs {
    Camera cam;
    uint32_t width, height, samples;
    uint64_t seed;
    uint64_t pixel_begin, pixel_count;
    float *fb;       // [3][pixel_count]
    uint8_t *fb_u8;  // [pixel_count][3] or null
};

#ifndef APT_Q8_POOL
#define APT_Q8_POOL 64 // 64 entries (2 KB): measured against 128 in round 3 -- the LDS it frees is worth more occupancy (C5 -3 %)
#endif
// Ray pool entries per wave = rays generated at a time (one batch fills the empty pool).  The 8-sphere form: 64, one per lane (ray-generate is
// a quarter of that kernel).  The grid form: 32 -- ray-generate is 2 % of a frame there, a half-empty generate pass costs nothing
// measurable, and the 1024 bytes of LDS it saves are a granule of 1280 bytes (round 4: 7 -> 6 granules; with round 5's half-pixel colour units 5).
__host__ __device__ constexpr uint32_t queue_pool_entries(bool grid) { return grid ? 32u : (uint32_t)APT_Q8_POOL; }
struct QueueArgs {
    uint32_t ppw;        // pixels per wave
    uint32_t nbuf;       // colour buffers = units that may be in flight (>= 2)
    uint32_t buf_bytes;  // bytes per colour buffer: queue_buf_bytes(maxleaf)
    uint32_t retire;     // grid form: 1 = APT_FLAG_RETIRE (finished paths stop), 0 = every path traces `depth` segments
};
// A colour buffer holds the 4 * n items of a unit as [sub-pixel][sample][3] floats, every sub-pixel's block shifted by 4 more
// bytes (a block of n * 12 bytes is a multiple of 256 bytes for n = 64: the four chains of a lane group would hit the same banks).
// Sub-pixels per unit.  The 8-sphere form: 4 (a unit = one pairwise leaf of one PIXEL).  The grid form: 2 (a unit = one leaf of HALF a pixel:
// sub-pixels {0, 1}, then {2, 3}; APT_QUEUE_GRID_SUBS): three buffers of 1544 bytes instead of two of 3088 bring a wave from 6 to 5 LDS
// granules of 1280 bytes at S = 64, i.e. from 20 to 24 waves per CU together with an 80-register budget -- that kernel hides its latencies with
// resident waves and little else (profiles/r05_c4_occupancy_sweep.jsonl).  For the 8-sphere form half-pixel units were measured no faster in
// round 4 (its vector pipe is full; twice as many unit sums cost what the waves return): it keeps whole pixels.
#ifndef APT_QUEUE_GRID_SUBS
#define APT_QUEUE_GRID_SUBS 2
#endif
__host__ __device__ constexpr uint32_t queue_unit_subs(bool grid) { return grid ? (uint32_t)APT_QUEUE_GRID_SUBS : 4u; }
__host__ __device__ inline uint32_t queue_buf_bytes(uint32_t maxleaf, uint32_t subs) { return subs * maxleaf * 12u + 4u * subs; }
// LDS of render_frame_queue8_kernel (all of it dynamic, so that the ray pool sits at LDS address 0 and a pool entry's address
// needs no base added), in bytes from its start: pool_a | pool_b | scene table | camera | roulette keys | sub-pixel sums | stack | colours
// (8576 bytes at S = 64 without roulette in the 8-sphere form; the grid form: 6040.  gfx950 hands out LDS in granules of 1280 bytes -- 128 per CU --, so 7 granules = 18 waves per CU
// whatever is shaved off down to 7680 bytes: a round-4 layout of 8176 bytes -- 16-bit colour addresses in a third pool array, no stored
// bounce countdown, compact camera -- measured 16.24 against 16.06 ms at C2 with retirement, its two extra address instructions per
// refill bought nothing.  Only the compact camera is kept.)
constexpr uint32_t kQueueTabFloats4 = 16;                // 8 centres + 8 albedos (the grid form keeps its always-tested pair slots there)
__host__ __device__ constexpr uint32_t queue_lds_off_tab(uint32_t pool) { return 2u * pool * 16u; }
__host__ __device__ constexpr uint32_t queue_lds_off_cam(uint32_t pool) { return queue_lds_off_tab(pool) + kQueueTabFloats4 * 16u; }
__host__ __device__ constexpr uint32_t queue_lds_off_key(uint32_t pool) { return queue_lds_off_cam(pool) + (uint32_t)sizeof(CameraLite); }
// (with roulette: the HIGH words of the keys, 4 bytes per pool entry; the low word takes the place of the bounce countdown in pool_b, which
// starts at depth - 1 for every ray and is not stored then -- 9072 -> 8816 bytes at S = 64: 7 instead of 8 LDS granules, 18 instead of 16 waves per CU)
// (behind the keys, half-pixel units only: 48 bytes = the sums of a pixel's four sub-pixels, [sub][channel], where decode_color's last step
// reads them -- the sub-pixels of a pixel are summed in different units)
__host__ __device__ inline uint32_t queue_lds_off_subres(uint32_t pool, bool rr) { return queue_lds_off_key(pool) + (rr ? pool * 4u : 0u); }
__host__ __device__ inline uint32_t queue_lds_off_stack(uint32_t pool, bool rr, uint32_t subs) { return queue_lds_off_subres(pool, rr) + (subs == 4u ? 0u : 48u); }
__host__ __device__ inline uint32_t queue_lds_off_colq(uint32_t pool, bool rr, bool stack, uint32_t subs) {
    return queue_lds_off_stack(pool, rr, subs) + (stack ? (uint32_t)kMaxStack * 3u * 4u * 4u : 0u);
}
__host__ __device__ inline uint32_t queue_lds_bytes(uint32_t pool, bool rr, uint32_t nbuf, bool stack, uint32_t buf_bytes, uint32_t subs) {
    return queue_lds_off_colq(pool, rr, stack, subs) + nbuf * buf_bytes;
}

#ifndef APT_QUEUE8_WAVES
#define APT_QUEUE8_WAVES 5 // waves per SIMD the register budget of render_frame_queue8_kernel is set for (96 VGPRs; its 8.6 KB of LDS allow 18 waves per CU at S = 64)
#endif
#ifndef APT_QUEUE_GRID_WAVES
#define APT_QUEUE_GRID_WAVES 6 // the grid form (SC == kSceneGrid): 80 registers (14 spilled, outside the walk loop) for a sixth wave per SIMD; with the half-pixel colour units its LDS (5 granules) allows it: 20 -> 24 waves per CU
#endif
// RR: APT_FLAG_RR (the host picks the instantiation from ta.rr_start): without it the kernel carries no roulette key and no
// per-bounce test of the flag.
// SC == kScene8: the reference's 8 spheres (SGPR scene, fast bounce block).  SC == kSceneGrid: any scene through the pair-slot
// tables of the uniform grid (apt_render_params.accel), every lane at its own place of its own walk: run_grid() below.
// STATS (grid form only): the per-lane walk statistics (cells visited, candidates tested) behind apt_set_trace_counter -- two registers and
// two instructions per loop turn that a frame without a counter does not pay (the host picks the instantiation from ta.traced).
template <int MODE, bool RR, int SC = kScene8, bool STATS = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SC == kSceneGrid ? APT_QUEUE_GRID_WAVES : APT_QUEUE8_WAVES, SC == kSceneGrid ? APT_QUEUE_GRID_WAVES : APT_QUEUE8_WAVES))) void render_frame_queue8_kernel(const float *__restrict__ sph, FrameArgs fa, TraceArgs ta,
                                                                 LeafProg lp, QueueArgs qa) {
    extern __shared__ __align__(16) unsigned char qlds[];
    constexpr uint32_t kPool = queue_pool_entries(SC == kSceneGrid), kPoolBatch = kPool;
    constexpr uint32_t kSubs = queue_unit_subs(SC == kSceneGrid), kHalves = 4u / kSubs;   // sub-pixels per unit; units per (pixel, leaf)
    static_assert(kSubs == 4u || kSubs == 2u, "a unit is a pixel or half a pixel");
    // (an LDS pointer made from the integer offset: the dynamic region starts at LDS address 0 -- checked below --, and through `qlds` every table
    // address carried an add of the symbol's link-time value, a v_add_u32 with 0 per bounce)
    typedef __attribute__((address_space(3))) float4 lds_float4;
    float4 *tab = (float4 *)(lds_float4 *)(uintptr_t)queue_lds_off_tab(kPool);
    CameraLite &cam = *reinterpret_cast<CameraLite *>(qlds + queue_lds_off_cam(kPool));
    const uint32_t lane = threadIdx.x;
    // No static LDS in this kernel, so the dynamic region starts at LDS address 0 (tests/test_isa_hazards.py checks the kernel
    // descriptor's group_segment_fixed_size); a build that breaks this renders nothing rather than reading the wrong pool entries.
    if ((uint32_t)(uintptr_t)qlds != 0u) { report_status(ta, APT_DEV_LDS_BASE); return; }
    if (SC == kSceneGrid && !grid_queue_usable(ta)) {                  // wave-uniform: render_frame_kernel renders this frame (its grid_walk == 2) --
        if (ta.grid_walk == 3u) report_status(ta, APT_DEV_GRID_MISMATCH);   // or, under APT_FLAG_GRID_SLOTS, nobody does: the caller's promise did not hold
        return;
    }
    if (lane == 0) cam = camera_lite(fa.cam);
    Scene8 sc;
    Tab8 tab8{tab, tab + 8};
    if (SC == kScene8) tab8 = load_scene8<false>(sph, sc, tab); // ends with a barrier
    else { sc.planes = false; __syncthreads(); }
    const KeyConsts kc = make_key_consts(ta.eps);
    const bool fast_ok = eps_allows_rootkey(ta.eps);
    constexpr bool rr = RR;
    const uint32_t nleaves = lp.nleaves, S = fa.samples, H = fa.height;
    const uint32_t nbuf = qa.nbuf;

    float4 *pool_a = reinterpret_cast<float4 *>(qlds);                 // (ox, oy, dx, dy)
    float4 *pool_b = pool_a + kPool;                                   // (oz, dz, colour address, bounce countdown)
    uint32_t *pool_keyhi = reinterpret_cast<uint32_t *>(qlds + queue_lds_off_key(kPool));     // Russian-roulette key, high word (APT_FLAG_RR only; low word: pool_b[.].w)
    float *subres = reinterpret_cast<float *>(qlds + queue_lds_off_subres(kPool, rr));   // [4][3]: sums of the pixel's sub-pixels
    float *stack = reinterpret_cast<float *>(qlds + queue_lds_off_stack(kPool, rr, kSubs));     // [kMaxStack][3][4] when nleaves > 1
    const uint32_t colq_off = queue_lds_off_colq(kPool, rr, nleaves > 1, kSubs);
    unsigned char *colq = qlds + colq_off;                              // [nbuf][items][3] floats
    constexpr uint32_t qlds_base = 0u;                                  // LDS byte address of the dynamic region (checked above)
    __syncthreads();

    // this wave's pixels
    const uint64_t wb = (uint64_t)xcd_chunked_block<8>(blockIdx.x, gridDim.x) * qa.ppw;   // (XCD-aware: neighbouring pixels through one L2)
    const uint32_t npx = (uint32_t)min((uint64_t)qa.ppw, fa.pixel_count - wb);
    const uint32_t U = npx * nleaves * kHalves;                         // units of this wave
    const uint64_t q0 = fa.pixel_begin + wb;

    // ---- wave-uniform queue state -------------------------------------------------------------------------------
    uint32_t pool_head = 0, pool_level = 0;                             // FIFO ring: entries [head, head + level)
    uint32_t g_unit = 0, g_off = 0, g_leaf = 0, g_start = 0, g_px = 0, g_buf = 0, g_half = 0; // ray-generate cursor
    uint32_t g_pi = (uint32_t)(q0 / H), g_pj = (uint32_t)(q0 % H);
    uint32_t a_unit = 0, a_leaf = 0, a_px = 0, a_buf = 0, a_sp = 0, a_half = 0;     // accumulation cursor (units are summed in order)
    uint64_t active = 0, alive = 0;                                     // lanes with a running path / that has not hit the light
    uint32_t issued = 0, a_end = kSubs * lp.len(0);                     // items handed to lanes so far / the count at which the oldest unsummed unit ends
    uint32_t traced = 0, n_bounce_exec = 0, n_gen_exec = 0, n_exact = 0; // statistics

    // ---- per-lane path state ------------------------------------------------------------------------------------
    PathState s;
    path_init(s, 0.f, 0.f, 0.f, 0.f, 0.f, 1.f);
    f2 thr_xy = {1.0f, 1.0f};
    float thr_z = 1.0f;
    uint32_t caddr = qlds_base + colq_off;                              // LDS address of the item's colour in its unit's buffer
    uint32_t left = 0;                                                  // bounces the path may still do after the next one
    uint64_t key = 0;                                                   // Russian-roulette key of the running path

    const Gain3 gain = load_gain(sph, ta);

    // ---- ray-generate: the next <= 64 items of the current unit into the pool ---------------------------------------
    auto gen_batch = [&]() __attribute__((always_inline)) {
        ++n_gen_exec;
        const uint32_t nl = lp.len(g_leaf), items = kSubs * nl;
        const uint32_t nb = min(kPoolBatch, items - g_off);
        const bool on = lane < nb;
        const uint32_t i = g_off + (on ? lane : 0u);                    // item within the unit: (sub - first sub of the unit) * nl + k
        const uint32_t sl = kSubs == 4u ? (i >= nl ? 1u : 0u) + (i >= 2u * nl ? 1u : 0u) + (i >= 3u * nl ? 1u : 0u) : (i >= nl ? 1u : 0u);
        const uint32_t sub = sl + kSubs * g_half;                       // (whole-pixel units: g_half stays 0)
        // path = ((q0 + g_px) * 4 + sub) * S + g_start + k (gen_data.py:32-36) with k = i - sub * nl, i.e. a wave-uniform base plus a
        // 32-bit per-lane offset: the 64-bit index arithmetic and the generator state of the base run on the scalar unit, a lane adds
        // offset * stride (one 32 x 64 bit product) -- round 3 formed the 64-bit path index and its 64 x 64 bit product per lane.
        const uint64_t base = (q0 + g_px) * 4u * (uint64_t)S + g_start;
        const uint32_t off = i - sl * nl + sub * S;                     // = sub * S + k  (< 4 * S)
        const uint64_t path = base + off;
        double u1, u2;
        // (splitmix64(seed) is recomputed here, on the scalar unit, rather than kept in two scalar registers across the hot loop)
        path_uniforms_at(splitmix64(fa.seed) + base * kPathStride + (uint64_t)off * kPathStride, u1, u2);
        float rox, roy, roz, rdx, rdy, rdz;
        camera_ray(cam, fa.width, fa.height, g_pi, g_pj, sub >> 1, sub & 1u, u1, u2, rox, roy, roz, rdx, rdy, rdz);
        // (a batch is generated only into an EMPTY pool -- kPoolBatch == kPool --, so it always starts at entry 0 and the FIFO never wraps)
        static_assert(kPoolBatch == kPool, "gen_batch restarts the pool at entry 0");
        pool_head = 0;
        const uint32_t e = pool_level + lane;
        if (on) {
            pool_a[e] = make_float4(rox, roy, rdx, rdy);
            const uint32_t ca = qlds_base + colq_off + g_buf * qa.buf_bytes + i * 12u + sl * 4u;
            if (rr) {
                // what the roulette of this path hashes at shading bounce d is splitmix64(key + phi * (d + 1)), and splitmix64 starts by adding
                // phi: the entry carries key + phi * rr_start, the state one step before the path's first roulette (d = rr_start - 1), and
                // roulette() below advances it by phi per draw -- no 64-bit multiply per bounce
                const uint64_t k64 = rr_path_key(ta.seed, path) + 0x9E3779B97F4A7C15ull * (uint64_t)ta.rr_start;
                pool_b[e] = make_float4(roz, rdz, __uint_as_float(ca), __uint_as_float((uint32_t)k64));
                pool_keyhi[e] = (uint32_t)(k64 >> 32);
            } else {
                pool_b[e] = make_float4(roz, rdz, __uint_as_float(ca), __uint_as_float(ta.depth - 1u));
            }
        }
        pool_level += nb;
        g_off += nb;
        if (g_off == items) {                                           // next unit: (the other half of this leaf,) the next leaf of the pixel, or the next pixel
            g_off = 0; ++g_unit;
            if (++g_buf == nbuf) g_buf = 0;
            if (kHalves == 1u || ++g_half == kHalves) {
                g_half = 0; g_start += nl;
                if (++g_leaf == nleaves) {
                    g_leaf = 0; g_start = 0; ++g_px;
                    if (++g_pj == H) { g_pj = 0; ++g_pi; }
                }
            }
        }
        __syncthreads(); // one wave per workgroup: orders the pool writes before the reads of other lanes
    };

    // ---- sum the oldest unit (all of its items are parked) as np.mean does; decode after the pixel's last leaf ------
    auto accumulate_unit = [&]() __attribute__((always_inline)) {
        __syncthreads();
        const uint32_t nl = lp.len(a_leaf), nfull = nl & ~7u, nt = nl - nfull;
        const uint32_t sl = (lane >> 3) & (kSubs - 1u), j = lane & 7u; // sub-pixel within the unit; lanes beyond 8 * kSubs repeat the work of the first
        const uint32_t sub = sl + kSubs * a_half;                       // sub-pixel of the pixel (whole-pixel units: a_half stays 0)
        const float *col = reinterpret_cast<const float *>(colq + a_buf * qa.buf_bytes);
        const uint32_t base = (sl * nl + j) * 3u + sl;                  // the sub-pixel's block is shifted by 4 bytes per sub-pixel
        float acc[3];
        acc[0] = col[base] * gain.r; acc[1] = col[base + 1] * gain.g; acc[2] = col[base + 2] * gain.b; // render.cpp:194-196
        for (uint32_t i8 = 8; i8 < nfull; i8 += 8) {                    // numpy's chain r[j] += a[j + 8m]
            const uint32_t o = base + i8 * 3u;
            acc[0] = acc[0] + col[o] * gain.r; acc[1] = acc[1] + col[o + 1] * gain.g; acc[2] = acc[2] + col[o + 2] * gain.b;
        }
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {                                // ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7))
            float v = acc[ch];
            v = v + __shfl_xor(v, 1, 64);
            v = v + __shfl_xor(v, 2, 64);
            v = v + __shfl_xor(v, 4, 64);
            acc[ch] = v;
        }
        if (nt) {                                                       // res += a[i] for the n % 8 trailing samples, in order
            const uint32_t o = (sl * nl + nfull + (j < nt ? j : 0u)) * 3u + sl;
            const float cr = col[o] * gain.r, cg = col[o + 1] * gain.g, cb = col[o + 2] * gain.b;
            for (uint32_t t = 0; t < nt; ++t) {
                const int src = (int)((lane & ~7u) + t);
                acc[0] = acc[0] + __shfl(cr, src, 64);
                acc[1] = acc[1] + __shfl(cg, src, 64);
                acc[2] = acc[2] + __shfl(cb, src, 64);
            }
        }
        float res[3] = {acc[0], acc[1], acc[2]};
        const bool last_half = kHalves == 1u || a_half + 1u == kHalves; // (the units of one leaf run the same stack program from the same level, each on its own sub-pixels' slots)
        if (nleaves > 1) {                                              // pairwise(left) + pairwise(right), innermost first
            uint32_t sp = a_sp;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) stack[(sp * 3 + ch) * 4 + sub] = acc[ch]; // the lanes of a group hold equal values
            ++sp;
            for (uint32_t m = 0; m < lp.ncomb(a_leaf); ++m) {
                --sp;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const float x = stack[((sp - 1) * 3 + ch) * 4 + sub], y = stack[(sp * 3 + ch) * 4 + sub];
                    stack[((sp - 1) * 3 + ch) * 4 + sub] = x + y;
                }
            }
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) res[ch] = stack[ch * 4 + sub];
            if (last_half) a_sp = sp;
        }
        if (a_leaf + 1 == nleaves) {                                    // decode_color: data_visualization.py:36-57
            const float fs = (float)S;
            const uint64_t pl = wb + a_px;
            if (kSubs == 4u) {
                // The lanes of a group hold equal sums for all three channels: lane j of a group decodes channel j (j < 3; the others repeat
                // channel 0), so the divide, the float64 mean of the four sub-pixels and the clamp run ONCE per pixel instead of once per channel
                // (round 4: ~60 of the ~205 vector instructions a wave spends per pixel outside the bounces; C2 with retirement 16.0 -> 15.75 ms.
                // Going further -- the lower half of the wave summing (r, g) as packed pairs, the upper half b -- measured no faster.)
                const int src0 = (int)((lane & ~31u) + j);              // the lane with this channel in sub-pixel group 0 of this half of the wave
                const uint32_t ch = j < 3u ? j : 0u;
                const float r = j == 1u ? res[1] : (j == 2u ? res[2] : res[0]);
                const float mean = r / fs;                              // np.mean: float32 sum / count
                double a64 = 0.0;                                       // :38 sum_color = zeros (float64)
#pragma unroll
                for (int sq = 0; sq < 4; ++sq) a64 = a64 + (double)__shfl(mean, src0 + sq * 8, 64); // :41-45
                const double v = a64 / 4;                               // :46
                const double cl = v < 0 ? 0 : (v > 1 ? 1 : v);          // :54
                if (lane < 3u) {
                    fa.fb[(uint64_t)ch * fa.pixel_count + pl] = (float)cl;
                    if (fa.fb_u8) fa.fb_u8[pl * 3 + ch] = (uint8_t)(cl * 255); // :55-57 truncation
                }
                a_sp = 0;
            } else {
                // half-pixel units: this unit's sub-pixel sums go to LDS; after the pixel's last unit the same decode reads all four from there
                if (lane < 8u * kSubs && j < 3u) subres[sub * 3u + j] = j == 1u ? res[1] : (j == 2u ? res[2] : res[0]);
                if (last_half) {
                    __syncthreads();
                    const uint32_t ch = lane < 3u ? lane : 0u;
                    double a64 = 0.0;                                   // :38 sum_color = zeros (float64)
#pragma unroll
                    for (int sq = 0; sq < 4; ++sq) a64 = a64 + (double)(subres[sq * 3 + ch] / fs);   // np.mean: float32 sum / count; :41-45
                    const double v = a64 / 4;                           // :46
                    const double cl = v < 0 ? 0 : (v > 1 ? 1 : v);      // :54
                    if (lane < 3u) {
                        fa.fb[(uint64_t)ch * fa.pixel_count + pl] = (float)cl;
                        if (fa.fb_u8) fa.fb_u8[pl * 3 + ch] = (uint8_t)(cl * 255); // :55-57 truncation
                    }
                    a_sp = 0;
                }
            }
        }
        ++a_unit;
        if (++a_buf == nbuf) a_buf = 0;
        if (kHalves == 1u || ++a_half == kHalves) {
            a_half = 0;
            if (++a_leaf == nleaves) { a_leaf = 0; ++a_px; }
        }
        a_end += kSubs * lp.len(a_leaf);                                // (past the wave's last unit the value is never used)
        __syncthreads();
    };
    // Are all items of the oldest unsummed unit parked?  They are when every one of them has been handed to a lane (the pool is a
    // FIFO, so that is a count) and no running lane still holds one (its colour address lies in the unit's buffer).  Evaluated
    // only when the generator needs the buffer or at the end -- a few times per unit -- so parking itself needs no counter
    // (the first form of this kernel bumped an LDS counter per parking lane: same address, so serialised, every bounce).
    auto oldest_unit_parked = [&]() __attribute__((always_inline)) -> bool {
        if (issued < a_end) return false;
        return (active & __builtin_amdgcn_ballot_w64(caddr - (colq_off + a_buf * qa.buf_bytes) < qa.buf_bytes)) == 0;
    };

    // ---- after a bounce: finished paths park their throughput (the colour is throughput * gain, applied when summed) ----
    auto park = [&]() __attribute__((always_inline)) {
        uint64_t zero, at_depth, saved;
        uint32_t orbits;
        // throughput (0,0,0): the OR of the three bit patterns is +-0 (all of them are): one v_or3, one float compare
        asm("v_or3_b32 %0, %1, %2, %3" : "=v"(orbits) : "v"(thr_xy.x), "v"(thr_xy.y), "v"(thr_z));
        asm("v_cmp_eq_f32_e64 %0, 0, %1" : "=s"(zero) : "v"(orbits));
        // countdown: the borrow is the mask of the lanes that have just done their last bounce
        asm("v_sub_co_u32_e64 %0, %1, %0, 1" : "+v"(left), "=s"(at_depth));
        const uint64_t done = active & (~alive | zero | at_depth);
        // executed even when no lane is done (exec = 0 then: nothing happens), which keeps the control flow of the hot loop flat
        asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                     "ds_write_b32 %[ca], %[rx]\n\t"
                     "ds_write_b32 %[ca], %[ry] offset:4\n\t"
                     "ds_write_b32 %[ca], %[rz] offset:8\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [sv] "=&s"(saved)
                     : [m] "s"(done), [ca] "v"(caddr), [rx] "v"(thr_xy.x), [ry] "v"(thr_xy.y), [rz] "v"(thr_z)
                     : "scc", "memory");
        active &= ~done;
    };
    // ---- idle lanes take the next pool entries, in pool order, straight into the state registers `st` --------------------
    auto refill = [&](PathState &st) __attribute__((always_inline)) {
        const uint64_t want = ~active;
        {   // no branch around this either: with nothing wanted or an empty pool `take` is 0
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(want >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)want, 0u));
            const uint64_t take = want & __builtin_amdgcn_ballot_w64(rank < pool_level);
            uint32_t ea;   // ((head + rank) * 16) mod 2048: one v_lshl_add_u32 with the scalar head * 16, one v_and
            asm("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(ea) : "v"(rank), "s"(pool_head << 4));
            uint64_t saved;
            if (!rr) {
                asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                             "ds_read_b64 %[oxy], %[ea]\n\t"
                             "ds_read_b64 %[dxy], %[ea] offset:8\n\t"
                             "ds_read_b32 %[oz], %[ea] offset:%[ob0]\n\t"
                             "ds_read_b32 %[dz], %[ea] offset:%[ob1]\n\t"
                             "ds_read_b32 %[ca], %[ea] offset:%[ob2]\n\t"
                             "ds_read_b32 %[lf], %[ea] offset:%[ob3]\n\t"
                             "v_pk_add_f32 %[rxy], 1.0, 0 op_sel_hi:[0,0]\n\t"
                             "v_mov_b32 %[rz], 1.0\n\t"
                             "s_waitcnt lgkmcnt(0)\n\t"
                             "s_mov_b64 exec, %[sv]"
                             : [sv] "=&s"(saved), [oxy] "+v"(st.oxy), [dxy] "+v"(st.dxy), [oz] "+v"(st.oz), [dz] "+v"(st.dz), [ca] "+v"(caddr),
                               [lf] "+v"(left), [rxy] "+v"(thr_xy), [rz] "+v"(thr_z)
                             : [m] "s"(take), [ea] "v"(ea), [ob0] "n"(kPool * 16u), [ob1] "n"(kPool * 16u + 4u), [ob2] "n"(kPool * 16u + 8u), [ob3] "n"(kPool * 16u + 12u)
                             : "scc", "memory");
            } else {
                // with roulette: the entry's fourth word is the key's low half, its high half comes from pool_keyhi (4-byte entries: address
                // ea / 4), the countdown starts at depth - 1 (wave-uniform)
                uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
                const uint32_t eh = ea >> 2;
                asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                             "ds_read_b64 %[oxy], %[ea]\n\t"
                             "ds_read_b64 %[dxy], %[ea] offset:8\n\t"
                             "ds_read_b32 %[oz], %[ea] offset:%[ob0]\n\t"
                             "ds_read_b32 %[dz], %[ea] offset:%[ob1]\n\t"
                             "ds_read_b32 %[ca], %[ea] offset:%[ob2]\n\t"
                             "ds_read_b32 %[kl], %[ea] offset:%[ob3]\n\t"
                             "ds_read_b32 %[kh], %[eh] offset:%[okh]\n\t"
                             "v_mov_b32 %[lf], %[l0]\n\t"
                             "v_pk_add_f32 %[rxy], 1.0, 0 op_sel_hi:[0,0]\n\t"
                             "v_mov_b32 %[rz], 1.0\n\t"
                             "s_waitcnt lgkmcnt(0)\n\t"
                             "s_mov_b64 exec, %[sv]"
                             : [sv] "=&s"(saved), [oxy] "+v"(st.oxy), [dxy] "+v"(st.dxy), [oz] "+v"(st.oz), [dz] "+v"(st.dz), [ca] "+v"(caddr),
                               [lf] "+v"(left), [rxy] "+v"(thr_xy), [rz] "+v"(thr_z), [kl] "+v"(klo), [kh] "+v"(khi)
                             : [m] "s"(take), [ea] "v"(ea), [eh] "v"(eh), [l0] "s"(ta.depth - 1u), [ob0] "n"(kPool * 16u), [ob1] "n"(kPool * 16u + 4u),
                               [ob2] "n"(kPool * 16u + 8u), [ob3] "n"(kPool * 16u + 12u), [okh] "n"(queue_lds_off_key(kPool))
                             : "scc", "memory");
                key = ((uint64_t)khi << 32) | klo;
            }
            const uint32_t nt = min((uint32_t)__popcll(want), pool_level);
            pool_head += nt;
            pool_level -= nt;
            issued += nt;
            active |= take;
            alive |= take;
        }
    };
    // ---- Russian roulette (include/render_mi355x.h APT_FLAG_RR; pt_core.h russian_roulette() is the per-lane form) for the lanes of `m`, which
    // have just shaded bounce d = depth - 1 - left (the countdown started at depth - 1).  Wave-level form: which lanes draw is a scalar mask
    // (d + 1 >= rr_start, alive, largest throughput component > 0), everything else runs unmasked -- an instruction costs its issue slot
    // whatever the mask -- and only the two writes are exec-masked: no nested divergent branches (the per-lane form compiled to four levels
    // of them), the hash input by one 64-bit add from the running state (see gen_batch), max3 for the largest component, the 24-bit draw
    // by a 32-bit conversion: ~35 instead of ~55 vector instructions per loop turn, same operations on the same values.
    auto roulette = [&](uint64_t m) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
        const int deep_left = (int)ta.depth - (int)ta.rr_start;         // d + 1 >= rr_start  <=>  left <= depth - rr_start
        const uint64_t m1 = m & __builtin_amdgcn_ballot_w64((int)left <= deep_left);
        if (m1 == 0) return;                                            // wave-uniform
        uint64_t saved;
        asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                     "v_lshl_add_u64 %[k], %[k], 0, %[phi]\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [sv] "=&s"(saved), [k] "+v"(key)
                     : [m] "s"(m1), [phi] "s"(0x9E3779B97F4A7C15ull)
                     : "scc");
        uint64_t x = key;                                               // splitmix64 after its first addition
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        const uint32_t hi = (uint32_t)(x >> 32);
        const uint32_t draw = (hi ^ (hi >> 31)) >> 8;                   // the 24 high bits of x ^ (x >> 31)
        float uf;
        asm("v_cvt_f32_u32 %0, %1" : "=v"(uf) : "v"(draw));
        const float u = uf * 0x1p-24f;
        // q = rx; if (ry > q) q = ry; if (rz > q) q = rz: the largest component with NaNs in ry / rz ignored -- v_max3 --, and NaN when rx is
        float q;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(q) : "v"(thr_xy.x), "v"(thr_xy.y), "v"(thr_z));
        const uint64_t draws = m1 & alive & __builtin_amdgcn_ballot_w64(q > 0.0f) & __builtin_amdgcn_ballot_w64(thr_xy.x == thr_xy.x);
        const float p = __builtin_amdgcn_fmed3f(q, 0.05f, 0.95f);
        const uint64_t dies = __builtin_amdgcn_ballot_w64(u >= p);
        const float r0 = __builtin_amdgcn_rcpf(p);                      // 1 / p, correctly rounded (russian_roulette's sequence)
        const float e0 = __builtin_fmaf(-p, r0, 1.0f);
        const float r1 = __builtin_fmaf(e0, r0, r0);
        float e = __builtin_fmaf(-p, r1, 1.0f);
        const float q1 = __builtin_fmaf(e, r1, r1);
        e = __builtin_fmaf(-p, q1, 1.0f);
        const float inv = __builtin_fmaf(e, r1, q1);
        const uint64_t lives = draws & ~dies, killed = draws & dies;
        f2 inv2;                                                        // (the packed product reads the first half twice: op_sel_hi)
        inv2.x = inv;
        asm volatile("s_and_saveexec_b64 %[sv], %[ml]\n\t"
                     "v_pk_mul_f32 %[rxy], %[rxy], %[inv] op_sel_hi:[1,0]\n\t"
                     "v_mul_f32 %[rz], %[rz], %[inv1]\n\t"
                     "s_mov_b64 exec, %[mk]\n\t"
                     "v_mov_b64 %[rxy], 0\n\t"
                     "v_mov_b32 %[rz], 0\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [sv] "=&s"(saved), [rxy] "+v"(thr_xy), [rz] "+v"(thr_z)
                     : [ml] "s"(lives), [mk] "s"(killed), [inv] "v"(inv2), [inv1] "v"(inv)
                     : "scc");
#endif
    };
    // ---- bookkeeping after a bounce that stands ------------------------------------------------------------------
    auto post_bounce = [&]() __attribute__((always_inline)) {
        ++n_bounce_exec;
        traced += (uint32_t)__popcll(active);
        if (rr) roulette(active);
    };
    // One bounce of the wave, in place.  Idle lanes compute on stale state: whatever they hold is overwritten when they take
    // their next ray.  The bounce runs in two phases (pt_trace.h bounce_ns8_v2_hit / _reflect): whether a lane whose path can still
    // reach an output left the validity range of the fast sequences (about 1e-5 of the wave-bounces) is known BEFORE the new ray
    // is written, so the exact form (sqrtf() and '/', a cold block) still finds the old ray in the state registers and the fast form
    // writes the new ray over them (fewer live registers and copies: C2 with retirement 16.69 -> 16.33 ms, depth 32 35.2 -> 34.4).

    // Service between two bounces: park, refill from what the pool holds; when the pool has room for a batch, (sum the oldest
    // unit if its buffer is needed,) generate 64 rays and refill again, so that a lane never idles because the pool ran dry.
    // -> true when the wave is finished: no lane got a ray although everything was offered, i.e. every unit has been
    // generated, issued and parked.  (If no lane is active and the pool is empty, every generated item is parked, so the oldest
    // unit can be summed and its buffer reused: generation is never blocked in that state.)
    auto service = [&](PathState &st) __attribute__((always_inline)) -> bool {
        park();
        refill(st);
        if (__builtin_expect(pool_level <= kPool - kPoolBatch && g_unit < U, 0)) {   // (unlikely: what only this path keeps in SGPRs is what should spill)
            bool room = g_off != 0 || g_unit - a_unit < nbuf;           // a unit needs a free colour buffer to start
            if (!room && oldest_unit_parked()) { accumulate_unit(); room = true; }
            if (room) { gen_batch(); refill(st); }
        }
        return active == 0;
    };
    // The hot loop: one bounce per turn, in place (no ping-pong pair: the cold exact block of step() merges into the same registers).
    // `guard`: an upper bound of the loop turns a wave can need (every turn either bounces an active lane or issues rays),
    // so that a logic error can never leave a wave spinning on the GPU.
#ifdef APT_TEST_TINY_GUARD   // test build only (tests/test_gpu_boundary.py): the bound trips at once, the status word must say so
    uint32_t guard = 3u;
#else
    uint32_t guard = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)min(4ull * ((uint64_t)npx * 4u * S * ((uint64_t)ta.depth + 1u) + 64u), 0xfffffffeull));
#endif
    // A tripped bound is reported AFTER the loop, from the counter itself (`guard-- == 0u` leaves it at 0xffffffff, which no other exit can):
    // the status pointer must not be live across the hot loop (it was, in the first form of this: 9-26 more spilled scalar registers, C2 with
    // retirement 16.1 -> 17.7 ms, C4 180 -> 204).
    constexpr uint32_t kGuardTripped = 0xffffffffu;
    auto run = [&](auto planes_tag) __attribute__((always_inline)) {
        {
            // The hot loop holds the FAST form only and leaves through an exit taken before anything of the new ray is written (phase 1
            // of the bounce decides): no join of a fast and an exact arm inside the loop, so the register allocator keeps the new ray in
            // the state registers.  The exact form (cold) runs between two visits of the hot loop.
            constexpr bool PLANES = decltype(planes_tag)::value;
            bool finished = false;
            while (!finished) {
                bool need_exact = false;
                for (;;) {
                    if (service(s) || guard-- == 0u) { finished = true; break; }
                    Bounce8Mid mid;
                    bool redo_any = !fast_ok;
                    if (__builtin_expect(fast_ok, 1)) {
                        const uint64_t redo = bounce_ns8_v2_hit<MODE, PLANES>(sc, tab8, s, ta, kc, mid) & active;
                        if (__builtin_expect(redo != 0, 0)) { // a finished path's request is ignored (trace_ns8)
                            const bool fin = select_const(alive, 1) == 0 || (thr_xy.x == 0.0f && thr_xy.y == 0.0f && thr_z == 0.0f);
                            redo_any = __builtin_amdgcn_ballot_w64(select_const(redo, 1) != 0 && !fin) != 0;
                        }
                    }
                    if (__builtin_expect(redo_any, 0)) { need_exact = true; break; }
                    Albedo albedo;
                    uint64_t alive_out = alive;
                    bounce_ns8_v2_reflect<MODE>(s, mid, alive_out, albedo);
                    apply_albedo(thr_xy, thr_z, albedo, alive_out);
                    alive = alive_out;
                    post_bounce();
                }
                if (need_exact) {
                    ++n_exact;
                    PathState c = s, o;
                    c.rxy = thr_xy; c.rz = thr_z; c.alive = select_const(alive, 1);
                    bounce_ns8_exact<MODE>(sc, tab8, c, o, ta);
                    s.oxy = o.oxy; s.oz = o.oz; s.dxy = o.dxy; s.dz = o.dz;
                    thr_xy = o.rxy; thr_z = o.rz;
                    alive = __builtin_amdgcn_ballot_w64(o.alive != 0);
                    post_bounce();
                }
            }
        }
    };
    // ---- SC == kSceneGrid: any scene through the grid's pair-slot tables, every lane at its own place of its own walk -----------
    // Why this form.  The nested walk of render_frame_kernel (pt_trace.h grid_segment) is bound by the CU's vector-memory ADDRESS
    // path, not by arithmetic or latency: its TA is 90-96 % busy (profiles/history/r03_grid_ta_pmc.json), and a wave-level load costs that
    // unit the same ~7 (dword) / ~17 (dwordx4) cycles whether 3 or 8 of its lanes are active (profiles/microbench/ta_rates.hip:
    // the cost only grows beyond ~8 lanes, by ~2.2 cycles per L1-missing lane).  The nested form issues ~320 such loads per wave
    // and segment, its float4 candidate loads serving 3.5 lanes on average: the inner loop runs to the longest list of the wave in
    // every cell of the longest walk, and the lanes of finished walks idle until the slowest of 64 is through.
    // Here a lane is WALKING or WAITING.  Per loop turn a walking lane tests ONE pair slot of its cell, if it has one left -- two candidates
    // in the packed form of the 8-sphere kernel (intersect_pre2, exact single-rsq square root, integer root keys) -- and then, its list
    // exhausted, leaves the cell (exit test, one DDA step, ONE dword with the next cell's slot range; a step out of the grid reads the
    // border layer's "outside" entry).  A lane whose walk has ended waits; as soon as `refill_lanes` lanes wait (or nobody walks) the
    // per-segment block runs for all of them together: shading step, roulette, park / refill from the ray pool, always-tested spheres
    // (through shared planes when they are the reference room), DDA set-up.
    // The arg-min carries (root key, position of the candidate): equal keys inside a list resolve by order (ids ascend there); a tie
    // with the minimum of an earlier list is recorded and settled by the two sphere ids when the segment is shaded (test_pair below).
    // Exactness: operation for operation intersect_pre / correctly rounded square root / select_root; a discriminant outside the
    // fast square root's range (|disc| < 2^-96) is redone with sqrtf() on the spot.  Lanes whose direction is not of unit length
    // test every sphere (trace_grid's rule).  The frame is bit-identical to render_frame_kernel's (tests/test_gpu_parity.py).
    auto run_grid = [&]() __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__) // (device-only builtins below)
        // The header and the always-tested list are wave-uniform, but the compiler cannot prove that the kernel's own stores never touch
        // the grid buffer, so it would fetch every header field with a VECTOR load (and a wait) at each use -- one per cell step for
        // `margin` alone, ~20 per segment start (C4: 226 -> 197 ms).  Explicit scalar loads: the header once (load_grid_header: scalar
        // registers, some of which live in spill lanes: a v_readlane per use), the list's slots where they are used.
        typedef float f32x8 __attribute__((ext_vector_type(8)));
        const GridHeader h = load_grid_header(ta.grid);
        const uint32_t *__restrict__ grid = ta.grid;
        const uint32_t ns = ta.ns;
        const uint32_t *cell_start = grid + h.off_cells, *cellslot = grid + h.off_cellslot, *slot_ids = grid + h.off_slot_ids;
        const float4 *geom = reinterpret_cast<const float4 *>(grid + h.off_geom);
        const float4 *slot_geom = reinterpret_cast<const float4 *>(grid + h.off_slots);
        const float4 *sphere8 = reinterpret_cast<const float4 *>(grid + h.off_sphere8);
        const int miss = (MODE == kModeOracle) ? -1 : 0;
        // The first kBigLds pair slots of the always-tested list live in LDS (the 8-sphere table's place, unused in this form): every segment
        // start reads them with uniform-address ds_read_b128 broadcasts, all issued before the first test.  Round 3 fetched each slot with a
        // scalar load and waited for it, four dependent memory latencies per segment start: re-running the list (measurement build,
        // profiles/r04_grid_ablation.jsonl) showed that one pass cost 46 % of the C4 frame, five times its instruction share.  Pairs beyond
        // kBigLds (more than 8 large spheres) keep the scalar loads.  Unused LDS pairs hold NaN spheres: a NaN discriminant is never a hit.
        constexpr uint32_t kBigLds = 4;
        static_assert(2 * kBigLds <= kQueueTabFloats4, "the LDS copy uses the 8-sphere table's region");
        if (lane < 2u * kBigLds) {
            const float qn = __uint_as_float(0x7fc00000u);
            tab[lane] = lane < 2u * min(h.slot_base, kBigLds) ? slot_geom[lane] : make_float4(qn, qn, qn, qn);
        }
        // What the DDA set-up reads of the header -- box corner, cell sizes and counts -- lives in LDS as well (four float4 behind the pair
        // slots): the per-segment block fetches them with uniform-address reads instead of holding fifteen more scalar registers across the
        // walk (this kernel's scalar registers are oversubscribed; which ones the allocator spills decided between 45 and 53 ms at C4 / 64 spp
        // for the same source, round 4).
        float4 *hl = tab + 2 * kBigLds;
        static_assert(2 * kBigLds + 4 <= kQueueTabFloats4, "header block behind the always-tested pair slots");
        if (lane == 0) {
            hl[0] = make_float4(h.gmin[0], h.gmin[1], h.gmin[2], h.margin);
            hl[1] = make_float4(h.inv_cell[0], h.inv_cell[1], h.inv_cell[2], __uint_as_float(h.n[0]));
            hl[2] = make_float4(h.cell[0], h.cell[1], h.cell[2], __uint_as_float(h.n[1]));
            hl[3] = make_float4(h.gmax[0], h.gmax[1], h.gmax[2], __uint_as_float(h.n[2]));
        }
        const uint32_t nbig = h.slot_base;                      // pair slots of the always-tested list
        const float walk_margin = h.margin;
        __syncthreads();
        // The generated scenes' always-tested list is the reference room: six walls (spheres 0..5 of the 8-sphere table, same order) and the
        // light, i.e. the equality pattern of scene8_shares_planes() with the light in the pad's place of the fourth pair.  A wave that finds
        // exactly that (bit for bit, on the LDS copy) replaces the copy by the eleven register pairs of intersect_pre_planes_pairs() -- the
        // fourth pair as (pad, light), positions swapped back when the arg-min records a hit -- and computes the eight discriminants in 49
        // instead of 64 packed instructions per segment start.
        bool planes7 = false;
        if (nbig == kBigLds) {
            float4 g[2 * kBigLds];
#pragma unroll
            for (uint32_t j = 0; j < 2u * kBigLds; ++j) g[j] = tab[j];
            auto eq = [](float a, float b) { return f32_bits(a) == f32_bits(b); };
            const bool ok = eq(g[2].x, g[2].y) && eq(g[2].x, g[4].x) && eq(g[2].x, g[4].y) && eq(g[2].x, g[6].x) &&      // cx2 = cx3 = cx4 = cx5 = cx(light)
                            eq(g[0].z, g[0].w) && eq(g[0].z, g[2].z) && eq(g[0].z, g[2].w) &&                            // cy0 = cy1 = cy2 = cy3
                            eq(g[1].x, g[1].y) && eq(g[1].x, g[5].x) && eq(g[1].x, g[5].y) && eq(g[1].x, g[7].x) &&      // cz0 = cz1 = cz4 = cz5 = cz(light)
                            g[6].y != g[6].y;                                                                              // position 7 is the pad
            planes7 = __builtin_amdgcn_readfirstlane(ok ? 1 : 0) != 0;
            if (planes7) {
                __syncthreads();
                if (lane == 0) {
                    const float qn = __uint_as_float(0x7fc00000u);
                    f2 *pp = reinterpret_cast<f2 *>(tab);
                    pp[0] = f2{g[0].x, g[0].y}; pp[1] = f2{g[2].x, qn};       // X1 = (cx0, cx1)   X2 = (cx2, pad)
                    pp[2] = f2{g[4].z, g[4].w}; pp[3] = f2{qn, g[6].z};       // Y1 = (cy4, cy5)   Y2 = (pad, cy light)
                    pp[4] = f2{g[0].z, g[0].z};                               // Y3 = (cy0, cy0)
                    pp[5] = f2{g[3].x, g[3].y}; pp[6] = f2{g[1].x, qn};       // Z1 = (cz2, cz3)   Z2 = (cz0, pad)
                    pp[7] = f2{g[1].z, g[1].w}; pp[8] = f2{g[3].z, g[3].w}; pp[9] = f2{g[5].z, g[5].w}; pp[10] = f2{qn, g[7].z};   // r2 pairs, the last (pad, light)
                }
                __syncthreads();
            }
        }
        // root keys (pt_trace.h KeyConsts), wave-uniform here: scalar registers
        const uint32_t kbias = f32_bits(ta.eps) + 1u, kinit = f32_bits(kMissT) - kbias;
        const uint64_t nbias2 = ((uint64_t)(0u - kbias) << 32) | (0x80000000u - kbias);
        const uint64_t retire_mask = qa.retire ? ~0ull : 0ull;
        // per-lane state of the running segment
        uint32_t bestk = 0, bestp = 0;                          // root key of the nearest accepted root so far; where its sphere is: position 2 * slot + half in
                                                                // slot_ids, or kIdFlag | sphere index (lanes that tested every sphere)
        constexpr uint32_t kIdFlag = 0x80000000u, kNoPos = 0xffffffffu;
        uint32_t pend = kNoPos;                                 // position of a candidate that tied with the running minimum (see test_pair)
        float tm0 = 0.f, tm1 = 0.f, tm2 = 0.f, td0 = 0.f, td1 = 0.f, td2 = 0.f;
        int inc0 = 0, inc1 = 0, inc2 = 0;                       // linear cell index increment per axis step
        uint32_t lin = 0, cur = 0, end = 0;                     // cell (BYTE offset into the BORDERED cellslot table), POSITION cursor / end (2 per slot)
        uint32_t n_cells = 0, n_tests = 0;                      // statistics
        uint64_t walking = 0, has_seg = 0;                      // lanes in a walk / lanes whose registers hold a (finished or running) segment
        // a wave mask (scalar) as the per-lane condition of a divergent region: the mask itself becomes the execution mask
        // (llvm.amdgcn.inverse.ballot; the v_cndmask + v_cmp pair of `select_const(m, 1) != 0` cost two vector instructions per use)
        auto lane_in = [&](uint64_t m) __attribute__((always_inline)) -> bool { return __builtin_amdgcn_inverse_ballot_w64(m); };
        // Two candidates (a pair slot) against the lane's ray; `pos` = position of the first.  Equal keys mean equal t: the lower SPHERE
        // index must win (the reference's strict '<' over ascending indices).  Inside a slot and inside a list the ids ascend, so an
        // earlier candidate beats a later one on a tie by order.  A tie with the running minimum from an EARLIER list -- nearly always
        // the same sphere met again in the next cell -- is only RECORDED (`pend`: position of the tying candidate; a strictly nearer
        // root voids it) and settled by the ids when the segment is shaded: no load, no wait in the walk.  A second tie while one is
        // recorded (rare: a sphere met in three cells) settles the recorded one on the spot.  (If both candidates tie with the running
        // minimum, the second has the higher id of the two and cannot matter.)
        auto id_at = [&](uint32_t p) __attribute__((always_inline)) -> uint32_t { return (p & kIdFlag) ? (p & ~kIdFlag) : slot_ids[p]; };
        // `ties_tag` false: the list is the segment's first (the always-tested one), there is no earlier minimum to tie with.
        // `swap_tag`: the pair's halves stand in the OTHER order than their positions (planes form, fourth pair): first half = position pos + 1
        auto test_post = [&](const HitPre2 hp, uint32_t pos, auto ties_tag, auto swap_tag) __attribute__((always_inline)) {
            constexpr bool TIES = decltype(ties_tag)::value;
            constexpr bool SWAP = decltype(swap_tag)::value;
            const f2 r0 = {__builtin_amdgcn_rsqf(hp.disc.x), __builtin_amdgcn_rsqf(hp.disc.y)};   // sqrt_rn_rsq1 on both halves
            const f2 y = hp.disc * r0, hh = r0 * 0.5f;
            const f2 res = __builtin_elementwise_fma(-y, y, hp.disc);
            f2 q = __builtin_elementwise_fma(res, hh, y);
            // NaN discriminants (pads, NaN spheres) drop out of the minimum: they cannot hit either way
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(min3_abs(1.0f, hp.disc.x, hp.disc.y) < 0x1p-96f) != 0, 0)) {
                asm volatile("" ::: "memory");
                q = f2{sqrtf(hp.disc.x), sqrtf(hp.disc.y)};
            }
            const uint64_t ka = root_pair<0>(hp.b, q) + nbias2, kb = root_pair<1>(hp.b, q) + nbias2;   // intersect_ns8_v2's keys
            const uint32_t ma = min((uint32_t)ka, (uint32_t)(ka >> 32)), mb = min((uint32_t)kb, (uint32_t)(kb >> 32));
            // The slot's own winner first (equal keys inside a slot: the first candidate, it has the lower id), then ONE comparison with the
            // running minimum: nearer -> take it (and void a recorded tie), equal -> record the tie, farther -> nothing.  (Round 3 compared
            // both candidates with the running minimum in turn: 4 more vector instructions per slot.)
            const bool isb = mb < ma;
            const uint32_t m2 = isb ? mb : ma;
            const uint32_t psel = SWAP ? (isb ? pos : pos + 1u) : (isb ? pos + 1u : pos);
            const bool take = m2 < bestk, tie = TIES && m2 == bestk;
            if (TIES && __builtin_expect(__builtin_amdgcn_ballot_w64(tie && pend != kNoPos) != 0, 0)) {
                if (tie && pend != kNoPos) {
                    if (slot_ids[pend] < id_at(bestp)) bestp = pend;
                    pend = kNoPos;
                }
            }
            bestp = take ? psel : bestp;
            if (TIES) pend = take ? kNoPos : (tie ? psel : pend);
            bestk = min(bestk, m2);                             // (last: the compares above read the old minimum)
        };
        auto test_pair = [&](const float4 a, const float4 c4, uint32_t pos, auto ties_tag) __attribute__((always_inline)) {
            test_post(intersect_pre2(a, c4, s.oxy.x, s.oxy.y, s.oz, s.dxy.x, s.dxy.y, s.dz), pos, ties_tag, std::false_type{});
        };
        // cellslot[lin] -> the cell's POSITION range [cur, end) in the pair-slot tables
        // a long list's slot count from cell_start (rare: clustered scenes); `lb`: bordered cell index -> the grid's own linear cell index
        auto long_count = [&](uint32_t lin4) __attribute__((always_inline)) -> uint32_t {
            const uint32_t lb = lin4 >> 2;
            const uint32_t n0 = __float_as_uint(hl[1].w), n1 = __float_as_uint(hl[2].w), sx = n0 + 2u, sy = n1 + 2u;
            const uint32_t xb = lb % sx, yb = (lb / sx) % sy, zb = lb / (sx * sy);
            const uint32_t c = ((zb - 1u) * n1 + (yb - 1u)) * n0 + (xb - 1u);
            return (cell_start[c + 1] - cell_start[c] + 1u) >> 1;
        };
        auto fetch_range = [&]() __attribute__((always_inline)) {
            const uint32_t cs = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(cellslot) + lin);
            cur = cs >> kGridSlotCountBits;                    // = 2 * slot_begin (kGridSlotShift)
            uint32_t cnt = cs & kGridSlotCountMax;
            if (cnt == kGridSlotCountMax) cnt = long_count(lin);   // a long list (clustered scenes)
            end = cur + 2u * cnt;
            if (STATS) ++n_cells;
        };
        // ---- one turn of the walking lanes ----
        // Written in wave masks: which lanes leave their cell, which stop, which test a slot are scalar values, the branches around the two
        // halves are wave-uniform and the per-lane updates are selects on those masks or exec-masked instructions.  (The first form put the
        // halves in divergent `if`s: every mask -> bool -> mask round trip is a v_cndmask + v_cmp pair, five of them per turn.)
        auto turn = [&]() __attribute__((always_inline)) {
            // FIRST the test of one pair slot for the lanes that have one, THEN the cell step for the lanes whose list is exhausted -- those of
            // empty cells and those that have just tested their last slot: the walk's last step (the one that finds "nothing nearer can lie
            // ahead") shares a turn with the last test instead of costing one of its own.  (Rounds 3-4 ran the halves in the other order: a
            // lane stepping into a non-empty cell tested its first slot in the same turn, and every walk ended with a turn that only stepped:
            // one turn more per segment whenever the first cell is not empty.)
            const uint64_t has = walking & __builtin_amdgcn_ballot_w64(cur < end);
            if (has != 0) {
                if (lane_in(has)) {
                    // slot cur / 2: float4s 2 * slot and 2 * slot + 1, by a 32-bit byte offset from the table's (scalar) base (positions have 27 bits)
                    const float4 *sg = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(slot_geom) + (cur << 4));
                    const float4 a = sg[0], c4 = sg[1];
                    test_pair(a, c4, cur, std::true_type{});
                    cur += 2u;
                    if (STATS) n_tests += 2;
                }
            }
            const uint64_t need = walking & __builtin_amdgcn_ballot_w64(cur >= end);     // list exhausted: leave the cell
            uint64_t stopm = 0;
            if (need != 0) {
                const float te = fminf(tm0, fminf(tm1, tm2));                           // parameter at which the ray leaves this cell
                const float tmin = bits_f32(bestk + kbias);
                // which axis is crossed: masks straight from the three compares
                const uint64_t b01 = __builtin_amdgcn_ballot_w64(tm0 <= tm1), b02 = __builtin_amdgcn_ballot_w64(tm0 <= tm2),
                               b12 = __builtin_amdgcn_ballot_w64(tm1 <= tm2);
                const uint64_t x0 = b01 & b02, m0 = x0 & need, m1 = ~x0 & b12 & need, m2 = need & ~(x0 | b12);
                // the crossed axis' boundary parameter and the cell index advance under the axis' own mask: three exec-masked groups of two
                // instructions (selects on the masks took nine: three sums, three selects for the parameters, two selects and a sum for the index)
                uint64_t saved;
                asm volatile("s_mov_b64 %[sv], exec\n\t"
                             "s_mov_b64 exec, %[m0]\n\t"
                             "v_add_f32 %[t0], %[t0], %[d0]\n\t"
                             "v_add_u32 %[lin], %[lin], %[i0]\n\t"
                             "s_mov_b64 exec, %[m1]\n\t"
                             "v_add_f32 %[t1], %[t1], %[d1]\n\t"
                             "v_add_u32 %[lin], %[lin], %[i1]\n\t"
                             "s_mov_b64 exec, %[m2]\n\t"
                             "v_add_f32 %[t2], %[t2], %[d2]\n\t"
                             "v_add_u32 %[lin], %[lin], %[i2]\n\t"
                             "s_mov_b64 exec, %[sv]"
                             : [sv] "=&s"(saved), [t0] "+v"(tm0), [t1] "+v"(tm1), [t2] "+v"(tm2), [lin] "+v"(lin)
                             : [m0] "s"(m0), [m1] "s"(m1), [m2] "s"(m2), [d0] "v"(td0), [d1] "v"(td1), [d2] "v"(td2), [i0] "v"(inc0), [i1] "v"(inc1), [i2] "v"(inc2));
                // nothing nearer can lie ahead
                // (te > 0 in a walk: te - (1e-3 * te + margin) as one fused multiply-add -- this is the walk's own conservative criterion, not
                // reference arithmetic; the margins are orders of magnitude above a rounding)
                stopm = need & __builtin_amdgcn_ballot_w64(tmin < __builtin_fmaf(te, 0.999f, -walk_margin));
                const uint64_t go = need & ~stopm;
                // cellslot[lin] -> the next cell's POSITION range (fetch_range()), for the lanes of `go`; the value is needed at once
                uint32_t cnt;
                asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                             "global_load_dword %[cnt], %[lin], %[base]\n\t"
                             "s_waitcnt vmcnt(0)\n\t"
                             "v_lshrrev_b32 %[cur], 6, %[cnt]\n\t"
                             "v_and_b32 %[cnt], 63, %[cnt]\n\t"
                             "v_lshl_add_u32 %[end], %[cnt], 1, %[cur]\n\t"
                             "s_mov_b64 exec, %[sv]"
                             : [sv] "=&s"(saved), [cur] "+v"(cur), [end] "+v"(end), [cnt] "=&v"(cnt)
                             : [m] "s"(go), [lin] "v"(lin), [base] "s"(cellslot)
                             : "scc", "memory");
                static_assert(kGridSlotCountBits == 6 && kGridSlotShift == 7 && kGridSlotCountMax == 63, "the shifts and masks of the block above");
                // ... or the step has left the grid: the border layer's entry (the count field's two top values: "outside" and "long list")
                const uint64_t special = go & __builtin_amdgcn_ballot_w64(cnt >= kGridCellOutside);
                if (special != 0) {
                    const uint64_t outside = special & __builtin_amdgcn_ballot_w64(cnt == kGridCellOutside), longl = special & ~outside;
                    stopm |= outside;
                    if (__builtin_expect(longl != 0, 0)) {       // a long list (clustered scenes): the count from cell_start
                        if (lane_in(longl)) end = cur + 2u * long_count(lin);
                    }
                }
                if (STATS) n_cells += select_const(go & ~stopm, 1);
                walking &= ~stopm;
            }
        };
        // ---- the per-segment block for the lanes of `batch` (none of them walking): finish the segment they hold, park / refill,
        // start the next one ----
        auto transition = [&](uint64_t batch) __attribute__((always_inline)) {
            ++n_bounce_exec;
            const uint64_t fin = batch & has_seg & active;     // lanes whose finished segment is shaded now
            traced += (uint32_t)__popcll(fin);
            bool is_light = false;                             // (wave masks are only formed outside divergent regions)
            Albedo alb;                                        // only read under exec = alive & fin (apply_albedo below): no default needed --
            asm("" : "=v"(alb.xy), "=v"(alb.z));                // "defined" for the compiler without an instruction
            if (lane_in(fin)) {
                const float tmin = bits_f32(bestk + kbias);
                // What the shading step needs of the winner: its centre -- straight from the winner's own pair slot, no sphere index needed --
                // and its index (light test, albedo), requested TOGETHER: one memory latency; the albedo (one more, it needs the index) is
                // only used after the reflection.  (Round 3: index -> sphere8 record -> shading, two latencies in front of the arithmetic.)
                const bool hit = bestk != kinit, by_pos = hit && !(bestp & kIdFlag);
                float cx, cy, cz;                               // set by exactly one of the two loads below (by_pos / !by_pos)
                asm("" : "=v"(cx), "=v"(cy), "=v"(cz));
                uint32_t id = bestp & ~kIdFlag;
                if (by_pos) {
                    // (32-bit byte offsets from the scalar table bases: position p -> word 8 * (p >> 1) + (p & 1) = 4 * p - 3 * (p & 1))
                    const float *sg = reinterpret_cast<const float *>(reinterpret_cast<const char *>(slot_geom) + ((bestp << 4) - 12u * (bestp & 1u)));
                    id = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(slot_ids) + (bestp << 2));
                    cx = sg[0]; cy = sg[2]; cz = sg[4];
                    if (__builtin_expect(pend != kNoPos, 0)) {   // a recorded tie: the lower index wins (nearly always the same sphere again)
                        const uint32_t idp = slot_ids[pend];
                        if (idp < id) {
                            const float *sp = reinterpret_cast<const float *>(slot_geom) + 8 * (size_t)(pend >> 1) + (pend & 1u);
                            id = idp;
                            cx = sp[0]; cy = sp[2]; cz = sp[4];
                        }
                    }
                }
                const int idx = hit ? (int)id : miss;
                const uint32_t g = (idx < 0) ? ns - 1 : (uint32_t)idx;
                if (!by_pos) { const float4 gcf = sphere8[2 * (size_t)g]; cx = gcf.x; cy = gcf.y; cz = gcf.z; }   // all-miss, or a lane that tested every sphere
                const float4 gc = make_float4(cx, cy, cz, 0.0f), col = sphere8[2 * (size_t)g + 1];   // albedo
                is_light = idx == ta.light;
                alb = Albedo{f2{col.x, col.y}, col.z};
                // GenerateNewRays in the packed form of the 8-sphere kernel's bounce (pt_trace.h reflect_packed: 2 operations per instruction for
                // the x / y components); outside the fast sequences' range the step is redone with sqrtf() and '/'
                PathState n;
                const float amin = reflect_packed<MODE>(s, tmin, gc.x, gc.y, gc.z, n);
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(amin >= kFastMin)) != 0, 0)) {
                    asm volatile("" ::: "memory");
                    PathState c = s;
                    c.rxy = thr_xy; c.rz = thr_z; c.alive = 1u;
                    shade_and_reflect<MODE>(c, tmin, gc.x, gc.y, gc.z, col.x, col.y, col.z, is_light);
                    n.oxy = c.oxy; n.oz = c.oz; n.dxy = c.dxy; n.dz = c.dz;
                }
                s.oxy = n.oxy; s.oz = n.oz; s.dxy = n.dxy; s.dz = n.dz;
            }
            // AccumulateIntervalColor (rt_helper.h:711-830): alive &= idx != light; ret *= alive ? albedo : 1 -- the product under exec = the
            // lanes of `fin` still alive after this bounce
            alive &= ~(fin & __builtin_amdgcn_ballot_w64(is_light));
            apply_albedo(thr_xy, thr_z, alb, alive & fin);
            if (rr) roulette(fin);
            {   // park the finished paths of `fin` (park() of the 8-sphere form, restricted to these lanes)
                uint64_t zero, at_depth, saved, zero_unused;
                uint32_t orbits;
                asm("v_or3_b32 %0, %1, %2, %3" : "=v"(orbits) : "v"(thr_xy.x), "v"(thr_xy.y), "v"(thr_z));
                asm("v_cmp_eq_f32_e64 %0, 0, %1" : "=s"(zero) : "v"(orbits));
                at_depth = __builtin_amdgcn_ballot_w64(left == 0u);
                asm("v_subbrev_co_u32_e64 %0, %1, 0, %0, %2" : "+v"(left), "=s"(zero_unused) : "s"(fin));   // left -= 1 in the lanes of fin (0 - 0 - borrow)
                const uint64_t done = fin & (((~alive | zero) & retire_mask) | at_depth);
                asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                             "ds_write_b32 %[ca], %[rx]\n\t"
                             "ds_write_b32 %[ca], %[ry] offset:4\n\t"
                             "ds_write_b32 %[ca], %[rz] offset:8\n\t"
                             "s_mov_b64 exec, %[sv]"
                             : [sv] "=&s"(saved)
                             : [m] "s"(done), [ca] "v"(caddr), [rx] "v"(thr_xy.x), [ry] "v"(thr_xy.y), [rz] "v"(thr_z)
                             : "scc", "memory");
                active &= ~done;
                has_seg &= ~done;
            }
            refill(s);
            if (__builtin_expect(pool_level <= kPool - kPoolBatch && g_unit < U, 0)) {
                bool room = g_off != 0 || g_unit - a_unit < nbuf;
                if (!room && oldest_unit_parked()) { accumulate_unit(); room = true; }
                if (room) { gen_batch(); refill(s); }
            }
            // ---- start the next segment of every active lane that is not walking ----
            const uint64_t start = active & ~walking;
            bool walk = false;
            if (lane_in(start)) {
                bestk = kinit; bestp = 0; pend = kNoPos;        // the key of kMissT: a root of exactly kMissT never wins, like the strict '<'
                cur = end = 0;
                const float dd = s.dxy.x * s.dxy.x + s.dxy.y * s.dxy.y + s.dz * s.dz;
                const bool unit = fabsf(dd - 1.0f) <= 1e-3f;   // false for NaN / inf
                if (!unit) {                                    // the roots are geometric ray parameters only for unit directions: every sphere,
                    float tmin = kMissT;                        // in the reference's own float form
                    uint32_t idx = 0;
                    for (uint32_t k = 0; k < ns; ++k) {
                        const float4 g = geom[k];
                        const float t = intersect_sphere(g.x, g.y, g.z, g.w, s.oxy.x, s.oxy.y, s.oz, s.dxy.x, s.dxy.y, s.dz, ta.eps);
                        if (t < tmin) { tmin = t; idx = k; }
                        if (STATS) ++n_tests;
                    }
                    bestk = f32_bits(tmin) - kbias;             // tmin is kMissT or an accepted root: its key is exact
                    bestp = kIdFlag | idx;
                } else {
                    {
                        // the always-tested list: its first kBigLds pair slots from LDS (uniform addresses: broadcasts, all in flight together;
                        // pads are NaN spheres), the rest -- scenes with more than 2 * kBigLds large spheres -- by scalar loads
                        // (Computing the four pairs' discriminants jointly in one basic block -- which saves the ten v_mov with which the compiler
                        // materialises the ray's broadcast pairs across the blocks of the pair tests -- measured 40.08 against 40.17 ms at 64 spp and
                        // cost 4 more spilled vector registers in the roulette instantiations: not kept.)
                        if (planes7) {                              // wave-uniform: the reference room (see above)
                            const f2 *pp = reinterpret_cast<const f2 *>(tab);
                            HitPre2 hp[4];
                            intersect_pre_planes_pairs(pp[0], pp[1], pp[2], pp[3], pp[4], pp[5], pp[6], pp[7], pp[8], pp[9], pp[10],
                                                       s.oxy.x, s.oxy.y, s.oz, s.dxy.x, s.dxy.y, s.dz, hp);
                            // (pinned: the compiler would sink parts of the joint arithmetic into the blocks of the four root tests and
                            // materialise the broadcast operands with v_mov there)
                            asm volatile("" : "+v"(hp[0].b), "+v"(hp[0].disc), "+v"(hp[1].b), "+v"(hp[1].disc), "+v"(hp[2].b), "+v"(hp[2].disc), "+v"(hp[3].b), "+v"(hp[3].disc));
                            test_post(hp[0], 0, std::false_type{}, std::false_type{});
                            test_post(hp[1], 2, std::false_type{}, std::false_type{});
                            test_post(hp[2], 4, std::false_type{}, std::false_type{});
                            test_post(hp[3], 6, std::false_type{}, std::true_type{});    // (pad, light): the light is position 6
                            if (STATS) n_tests += 8;
                        } else {
                            float4 big[2 * kBigLds];
#pragma unroll
                            for (uint32_t j = 0; j < 2u * kBigLds; ++j) big[j] = tab[j];
#pragma unroll
                            for (uint32_t j = 0; j < kBigLds; ++j) {
                                if (j < nbig) {                         // wave-uniform
                                    test_pair(big[2 * j], big[2 * j + 1], 2 * j, std::false_type{});
                                    if (STATS) n_tests += 2;
                                }
                            }
                        }
                        for (uint32_t j = kBigLds; j < nbig; ++j) {
                            f32x8 g8;
                            asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(g8) : "s"(slot_geom + 2 * j) : "memory");
                            test_pair(make_float4(g8[0], g8[1], g8[2], g8[3]), make_float4(g8[4], g8[5], g8[6], g8[7]), 2 * j, std::false_type{});
                            if (STATS) n_tests += 2;
                        }
                    }
                }
                if (unit) {
                    // DDA set-up.  The origin of every segment but a path's first lies on a sphere's surface, and in a closed room that is
                    // inside the grid box (so do the camera's ray origins in the demo scenes): the cell of the origin is where the walk
                    // starts, and whether the origin IS in the box comes out of the cell computation itself (0 <= cell < n, unsigned).  Only
                    // when some starting lane's origin lies outside (wave-level, rare) does the slab test decide where -- whether -- its ray
                    // enters the box (round 3 ran slab test, entry point and clamps for every segment: ~55 of this block's vector instructions).
                    const float ix = __builtin_amdgcn_rcpf(s.dxy.x), iy = __builtin_amdgcn_rcpf(s.dxy.y), iz = __builtin_amdgcn_rcpf(s.dz);
                    const float4 hg = hl[0], hi4 = hl[1], hc = hl[2], hx = hl[3];   // (gmin, margin) (1 / cell, n0) (cell, n1) (gmax, n2): LDS broadcasts
                    const int n0 = (int)__float_as_uint(hi4.w), n1 = (int)__float_as_uint(hc.w), n2 = (int)__float_as_uint(hx.w);
                    const float e0 = s.oxy.x - hg.x, e1 = s.oxy.y - hg.y, e2 = s.oz - hg.z;   // origin relative to the box corner
                    auto cell_of = [](float f) __attribute__((always_inline)) -> int {   // floor + convert in one instruction (saturating; NaN -> 0)
                        int c;
                        asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(c) : "v"(f));
                        return c;
                    };
                    int c0 = cell_of(e0 * hi4.x), c1 = cell_of(e1 * hi4.y), c2 = cell_of(e2 * hi4.z);
                    const bool inside = (uint32_t)c0 < (uint32_t)n0 && (uint32_t)c1 < (uint32_t)n1 && (uint32_t)c2 < (uint32_t)n2;
                    bool go = true;
                    float tn = 0.0f;
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!inside) != 0, 0)) {
                        asm volatile("" ::: "memory");
                        float tf = 3.0e38f;
                        bool inbox = true;
                        auto slab = [&](float o, float dv, float inv, float lo, float hi) __attribute__((always_inline)) {
                            if (fabsf(dv) > 1e-20f) {
                                const float t1 = (lo - o) * inv, t2 = (hi - o) * inv;
                                tn = fmaxf(tn, fminf(t1, t2));
                                tf = fminf(tf, fmaxf(t1, t2));
                            } else if (!(o >= lo && o <= hi)) inbox = false;
                        };
                        slab(s.oxy.x, s.dxy.x, ix, hg.x, hx.x);
                        slab(s.oxy.y, s.dxy.y, iy, hg.y, hx.y);
                        slab(s.oz, s.dz, iz, hg.z, hx.z);
                        go = inbox && tn <= tf;
                        auto entry_cell = [&](float e, float dv, float invw, int na) __attribute__((always_inline)) -> int {
                            const int ci = (int)floorf((e + dv * tn) * invw);      // (an origin inside has tn = 0: the cell computed above)
                            return ci < 0 ? 0 : (ci >= na ? na - 1 : ci);
                        };
                        c0 = entry_cell(e0, s.dxy.x, hi4.x, n0);
                        c1 = entry_cell(e1, s.dxy.y, hi4.y, n1);
                        c2 = entry_cell(e2, s.dz, hi4.z, n2);
                    }
                    if (go) {
                        // per axis: the parameter of the boundary ahead -- plane ci + 1 (dv > 0) or ci (dv < 0) of the axis, relative to the
                        // box corner like e --, the parameter step per cell (-cellw * inv = cellw * |inv| for dv < 0), the linear-index step
                        // in the bordered cellslot table.  An axis the ray does not move along gets a crossing that never comes first (it is never the
                        // nearest crossing of a unit direction within kMissT), so its other values are never used.
                        auto axis = [&](float e, float dv, float inv, float cellw, int ci, int stride, int &inc, float &tmax,
                                        float &tdel) __attribute__((always_inline)) {
                            const bool fwd = dv > 0.0f, moves = fabsf(dv) > 1e-20f;
                            const float t_b = __builtin_fmaf((float)(ci + (fwd ? 1 : 0)), cellw, -e) * inv;
                            tmax = moves ? t_b : 3.0e38f;
                            tdel = moves ? cellw * fabsf(inv) : 3.0e38f;
                            inc = fwd ? stride : -stride;
                        };
                        const int sx = n0 + 2, sy = n1 + 2;         // strides of the bordered table; `lin` and its increments are BYTE offsets
                        axis(e0, s.dxy.x, ix, hc.x, c0, 4, inc0, tm0, td0);
                        axis(e1, s.dxy.y, iy, hc.y, c1, 4 * sx, inc1, tm1, td1);
                        axis(e2, s.dz, iz, hc.z, c2, 4 * sx * sy, inc2, tm2, td2);
                        lin = (uint32_t)(((c2 + 1) * sy + (c1 + 1)) * sx + (c0 + 1)) << 2;
                        // (Requesting this first range before the axis arithmetic and waiting after it -- inline-asm load -- was measured:
                        // 52.5 against 44.7 ms at 64 spp; the asm's memory clobber doubled the spilled scalar registers.)
                        fetch_range();
                        walk = true;
                    }
                }
            }
            walking |= __builtin_amdgcn_ballot_w64(walk);
            has_seg |= start;
        };
        const uint32_t batch_lanes = ta.refill_lanes;          // waiting lanes that trigger the per-segment block (apt_set_refill_lanes; default 32)
        // upper bound of the loop turns (a turn advances a walking lane by a cell or a slot, or a transition issues / shades):
        // protection against a logic error, never reached
#ifdef APT_TEST_TINY_GUARD
        uint64_t turns_left = 3u;
#else
        uint64_t turns_left = ((uint64_t)npx * 4u * S * ((uint64_t)ta.depth + 1u) + 64u) * ((uint64_t)h.n[0] + h.n[1] + h.n[2] + 5u + h.nslots);
#endif
        for (;;) {
            const uint64_t waiting = ~walking;                  // finished segments, fresh lanes and lanes without a path
            const uint32_t nwait = (uint32_t)__popcll(waiting & active) + ((pool_level != 0 || g_unit < U) ? (uint32_t)__popcll(~active) : 0u);
            if (walking == 0 || nwait >= batch_lanes) {
                transition(waiting);
                if (active == 0) break;                         // nothing runs, nothing could be issued: everything is parked
            }
            if (turns_left-- == 0u) break;
            turn();
        }
        if (turns_left == ~0ull) report_status(ta, APT_DEV_GRID_TURNS);   // (only the bound's own exit leaves the counter there; see `guard`)
        if (STATS && ta.traced) {                               // statistics: cells visited / candidates tested (grid_stats)
            unsigned long long c = n_cells, t = n_tests;
            for (int off = 32; off > 0; off >>= 1) { c += __shfl_xor(c, off, 64); t += __shfl_xor(t, off, 64); }
            if (lane == 0) { atomicAdd(ta.traced + 1, c); atomicAdd(ta.traced + 2, t); }
        }
#endif
    };
    if (SC == kSceneGrid) run_grid();
    else if (sc.planes) run(std::true_type{});
    else run(std::false_type{});
    if (SC != kSceneGrid && guard == kGuardTripped) report_status(ta, APT_DEV_QUEUE_GUARD);
    // No lane is active, the pool is empty and every unit has been generated: every item is parked.  (After a tripped loop bound the
    // sums below read items that were never parked: the frame is incomplete and the status word says so.)
    while (a_unit < U) accumulate_unit();

    if (ta.traced && lane == 0) {
        atomicAdd(ta.traced, (unsigned long long)traced);
        if (SC != kSceneGrid) {                                 // (the grid form reports its walk statistics in these two slots)
            atomicAdd(ta.traced + 1, 64ull * n_bounce_exec);
            atomicAdd(ta.traced + 2, 64ull * n_gen_exec);
        }
        if (n_exact) atomicAdd(ta.traced + 3, (unsigned long long)n_exact);
    }
}

} // namespace
