// render_kernels.hip -- gfx950 (MI355X, wave64) kernels of the path-tracing hot path and
// the C-ABI of librender_mi355x.so (include/render_mi355x.h).
//
// One lane = one path; the whole bounce loop lives in registers (the reference moves 64-ray
// tiles through a 16 KB scratch buffer with a free-list allocator, src/allocator.h -- no
// counterpart here).  No MFMA: this is branchy fp32 VALU work.  Built with
// -ffp-contract=off (see pt_core.h) so that results are bitwise those of the CPU
// restatement; sqrt and divide are hipcc's correctly rounded expansions.
//
// Kernels
//   render_paths_kernel   rays from a [6][N] buffer -> colours [3][N]
//                         = src/render.cpp:40-60,82-223 (Process/CopyIn/Compute/CopyOut)
//   render_frame_kernel   ray-generate + trace + per-pixel accumulation, nothing
//                         materialised = gen_data.py:21-75 + render.cpp:104-207 +
//                         data_visualization.py:20-59
//   gen_rays_kernel       device gen_rays (counter RNG)
//   decode_color_kernel   device decode_color
//
// Scene placement
//   Ns == 8 (the reference scene): the 32 geometry floats are read once per wave with
//   scalar loads (wave-uniform addresses into a read-only buffer -> s_load into SGPRs) and
//   every intersect instruction takes its sphere operand from an SGPR; centre/albedo of the
//   hit sphere are gathered per lane from a 256-byte LDS table (8 distinct 16-byte slots,
//   conflict free).
//   Any Ns: [cx,cy,cz,r2] tiles of 1024 spheres are staged through LDS by the whole
//   workgroup (coalesced plane loads) and read back as wave-uniform ds_read_b128
//   broadcasts; the sqrt half of the intersection is skipped when no lane of the wave has a
//   non-negative discriminant.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/render_mi355x.h"
#include "pt_core.h"

namespace {

using namespace apt;

constexpr int kBlock = 256;      // 4 waves
constexpr int kScene8 = 0, kSceneTiles = 1, kSceneGrid = 2; // template parameter SC: how the scene reaches the lanes
constexpr int kTile = 1024;      // spheres per LDS tile (16 KB)
constexpr int kMaxLeaves = 64;   // pairwise-sum leaves -> samples <= 8192
#ifndef APT_GRID_WAVES
#define APT_GRID_WAVES 8 // min waves per SIMD requested for the grid-walk kernels: the walk is latency bound
                         // (dependent cell -> item loads), measured 464 / 371 / 334 / 311 / 302 ms at 3 / 4 / 5 / 6 / 8 waves
#endif
#ifndef APT_FULL_WAVES
#define APT_FULL_WAVES 1 // min waves per SIMD requested for the full-trace frame kernel (A/B knob)
#endif
constexpr int kMaxStack = 8;
constexpr int kStackSlots = kBlock / 8; // one pairwise-sum stack per sub-pixel group (its 8 lanes hold equal values)
constexpr uint32_t kRefillLanes = 32; // default: lanes with an empty ray slot that trigger a wave-wide ray-generate

struct Scene8 { // wave-uniform registers (SGPRs)
    float cx[8], cy[8], cz[8], r2[8];
};

struct TraceArgs {
    uint32_t ns;
    uint32_t depth;
    int32_t light;
    float eps, gain;
    uint32_t refill_lanes;      // compaction: batch size that triggers ray-generate (tuning knob)
    uint32_t emission;          // APT_FLAG_EMISSION: gain per channel = emission of sphere `light` instead of `gain`
    uint32_t rr_start;          // Russian roulette (APT_FLAG_RR): first bounce count it applies at; 0 = off
    uint64_t seed;              // keys the roulette draws
    const uint32_t *grid;       // apt_render_params.accel (device) or null
    unsigned long long *traced; // optional device counter of traced segments
};

struct LeafProg { // numpy pairwise_sum recursion flattened (see build_leaves)
    uint32_t nleaves;
    uint32_t maxleaf;          // longest leaf (sizes the refill colour queue)
    uint32_t leaf[kMaxLeaves]; // len | ncomb << 16 (dwords: wave-uniform s_load from the kernarg segment)
    __host__ __device__ uint32_t len(uint32_t i) const { return leaf[i] & 0xffffu; }
    __host__ __device__ uint32_t ncomb(uint32_t i) const { return leaf[i] >> 16; }
};

// Discriminants of TWO spheres per instruction: the tile is stored as sphere pairs,
//   tile[2p]   = (cx[2p], cx[2p+1], cy[2p], cy[2p+1])      tile[2p+1] = (cz[2p], cz[2p+1], r2[2p], r2[2p+1])
// so every operation of intersect_pre becomes one v_pk_{add,mul}_f32 over a register pair, the
// ray component being broadcast to both halves by op_sel (no register shuffles).  Packed fp32
// ops round exactly like the scalar ones, element by element; contraction is off.
typedef float f2 __attribute__((ext_vector_type(2)));
struct HitPre2 { f2 b, disc; };
__device__ __forceinline__ HitPre2 intersect_pre2(const f2 cx, const f2 cy, const f2 cz, const f2 r2, float ox,
                                                  float oy, float oz, float dx, float dy, float dz) {
    const f2 ocx = cx - ox, ocy = cy - oy, ocz = cz - oz;
    f2 b = ocx * dx;
    b = b + ocy * dy;
    b = b + ocz * dz;
    f2 c = ocx * ocx;
    c = c + ocy * ocy;
    c = c + ocz * ocz;
    c = c - r2;
    f2 disc = b * b;
    disc = disc - c;
    return {b, disc};
}
__device__ __forceinline__ HitPre2 intersect_pre2(const float4 a, const float4 c4, float ox, float oy, float oz,
                                                  float dx, float dy, float dz) {
    return intersect_pre2(f2{a.x, a.y}, f2{a.z, a.w}, f2{c4.x, c4.y}, f2{c4.z, c4.w}, ox, oy, oz, dx, dy, dz);
}


// ---- trace: reference scene (Ns == 8) ----------------------------------------------------
// One bounce: 8 intersections (sphere operands in SGPRs), arg-min, gather, shade.
// FAST: exact fast sqrt sequences (pt_core.h) and, when eps permits, the integer-key arg-min.
template <int MODE, bool FAST>
__device__ __forceinline__ bool bounce_ns8(const Scene8 &sc, const float4 *tab, const PathState &s, PathState &n,
                                           const TraceArgs &ta) {
    float amin = 1.0f; // min |sqrt argument| of this bounce (FAST only)
    float tmin;
    int idx;
    const int miss = (MODE == kModeOracle) ? -1 : 0; // all-miss: gen_data.py:311 / rt_helper.h:183-201
    if (FAST) {
        RootKey key;
        rootkey_init(key, ta.eps, miss);
#if defined(APT_NS8_SCALAR) // A/B switch: one sphere per scalar instruction stream
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float t0, t1;
            intersect_roots<true>(sc.cx[k], sc.cy[k], sc.cz[k], sc.r2[k], s.ox, s.oy, s.oz, s.dx, s.dy, s.dz, t0, t1,
                                  amin);
            rootkey_update(key, t0, t1, k);
        }
#else
#pragma unroll
        for (int k = 0; k < 8; k += 2) { // rt_helper.h:457-467, two spheres per packed instruction
            const HitPre2 h = intersect_pre2(f2{sc.cx[k], sc.cx[k + 1]}, f2{sc.cy[k], sc.cy[k + 1]},
                                             f2{sc.cz[k], sc.cz[k + 1]}, f2{sc.r2[k], sc.r2[k + 1]}, s.ox, s.oy, s.oz,
                                             s.dx, s.dy, s.dz);
            // sqrt_rn_rsq1 on both lanes of the pair (pt_core.h): y = x*r, hh = r/2, q = fma(fma(-y,y,x), hh, y)
            amin = fminf(amin, fminf(fabsf(h.disc.x), fabsf(h.disc.y)));
            const f2 r0 = {__builtin_amdgcn_rsqf(h.disc.x), __builtin_amdgcn_rsqf(h.disc.y)};
            const f2 y = h.disc * r0, hh = r0 * 0.5f;
            const f2 res = __builtin_elementwise_fma(-y, y, h.disc);
            const f2 q = __builtin_elementwise_fma(res, hh, y);
            const f2 t0 = h.b - q, t1 = h.b + q;
            rootkey_update(key, t0.x, t1.x, k);
            rootkey_update(key, t0.y, t1.y, k + 1);
        }
#endif
        tmin = rootkey_tmin(key);
        idx = key.idx;
    } else {
        tmin = kMissT;
        idx = miss;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float t0, t1;
            intersect_roots<false>(sc.cx[k], sc.cy[k], sc.cz[k], sc.r2[k], s.ox, s.oy, s.oz, s.dx, s.dy, s.dz, t0, t1,
                                   amin);
            const float t = select_root(t0, t1, ta.eps);
            if (t < tmin) { tmin = t; idx = k; } // strict '<', ascending k: lowest index wins ties
        }
    }
    const int g = (idx < 0) ? 7 : idx; // Python index -1 wraps to the last sphere
    const float4 c = tab[2 * g], col = tab[2 * g + 1];
    n = s;
    shade_and_reflect<MODE, FAST>(n, tmin, c.x, c.y, c.z, col.x, col.y, col.z, idx == ta.light, &amin);
    // the fast sequences are only valid for |sqrt argument| >= 2^-96 and 0 < eps < 1e20
    return FAST && (amin < 0x1p-96f || !eps_allows_rootkey(ta.eps));
}

template <int MODE, bool RETIRE>
__device__ __forceinline__ uint32_t trace_ns8(const Scene8 &sc, const float4 *tab, PathState &s, bool valid,
                                              const TraceArgs &ta, uint64_t path) {
    uint32_t traced = 0;
    const uint64_t rr_key = ta.rr_start ? rr_path_key(ta.seed, path) : 0;
    for (uint32_t d = 0; d < ta.depth; ++d) { // render.cpp:140-188
        const bool fin = RETIRE && (!valid || path_finished(s));
        if (RETIRE && __all(fin)) break;
        PathState n;
        bool redo = bounce_ns8<MODE, true>(sc, tab, s, n, ta);
        if (__builtin_expect(__any(redo), 0)) {
            // A lane left the validity range of the fast sequences (|sqrt argument| < 2^-96, divide
            // operands outside [2^-40, 2^40]).  A lane whose path is already finished (alive bit
            // cleared or throughput zero) cannot influence any output any more, so its request is
            // ignored: deep all-miss paths (|n| ~ 1e20) are of that kind.  Otherwise redo the bounce
            // with sqrtf() and '/'.  The empty volatile asm keeps this cold path out of the hot block.
            redo = redo && !path_finished(s);
            if (__any(redo)) {
                asm volatile("" ::: "memory");
                (void)bounce_ns8<MODE, false>(sc, tab, s, n, ta);
                if (ta.traced && (threadIdx.x & 63) == 0) atomicAdd(ta.traced + 3, 1ull); // statistics: exact re-runs
            }
        }
        if (ta.rr_start && d + 1 >= ta.rr_start) russian_roulette(n, rr_key, d); // wave-uniform branch
        if (RETIRE) {
            if (!fin) { s = n; ++traced; }
        } else { // full trace: lanes past the end of the range compute garbage that is never stored
            s = n;
            ++traced;
        }
    }
    return traced;
}

// ---- trace: any scene, LDS-staged tiles -------------------------------------------------
// Every thread of the workgroup must call this together (it contains barriers).
template <int MODE, bool RETIRE>
__device__ __forceinline__ uint32_t trace_dyn(const float *__restrict__ sph, float4 *tile, PathState &s, bool valid,
                                              const TraceArgs &ta, uint64_t path) {
    const uint64_t rr_key = ta.rr_start ? rr_path_key(ta.seed, path) : 0;
    const uint32_t ns = ta.ns;
    const float *r2 = sph, *cx = sph + ns, *cy = sph + 2 * (size_t)ns, *cz = sph + 3 * (size_t)ns;
    const float *colx = sph + 7 * (size_t)ns, *coly = sph + 8 * (size_t)ns, *colz = sph + 9 * (size_t)ns;
    uint32_t traced = 0;
    for (uint32_t d = 0; d < ta.depth; ++d) {
        const bool fin = !valid || (RETIRE && path_finished(s));
        if (RETIRE && __syncthreads_and(fin)) break;
        float tmin = kMissT;
        int idx = (MODE == kModeOracle) ? -1 : 0;
        for (uint32_t base = 0; base < ns; base += kTile) {
            const uint32_t n = min((uint32_t)kTile, ns - base);
            __syncthreads(); // previous tile fully consumed
            {   // stage: coalesced plane loads, pair-interleaved LDS layout, NaN spheres pad the tail to a
                // multiple of 4 (a NaN discriminant is never >= 0, so a pad can never hit)
                float *tf = reinterpret_cast<float *>(tile);
                const uint32_t n4 = (n + 3u) & ~3u;
                for (uint32_t k = threadIdx.x; k < n4; k += kBlock) {
                    const bool real = k < n;
                    const float qn = __uint_as_float(0x7fc00000u);
                    const uint32_t o = (k >> 1) * 8u + (k & 1u);
                    tf[o] = real ? cx[base + k] : qn;
                    tf[o + 2] = real ? cy[base + k] : qn;
                    tf[o + 4] = real ? cz[base + k] : qn;
                    tf[o + 6] = real ? r2[base + k] : qn;
                }
            }
            __syncthreads();
            // Four spheres per step: four wave-uniform ds_read_b128 broadcasts in flight together, two
            // packed discriminant evaluations, ONE test "can any lane hit any of the four?".  A
            // negative discriminant yields kMissT, which never wins the strict '<', so skipping the
            // sqrt/root half for misses is result preserving; hits are then taken in ascending
            // sphere order, which keeps the lowest-index-on-ties rule.
            auto hit = [&](float b, float disc, uint32_t sphere) {
                if (__any(disc >= 0.0f)) {
                    const float t = intersect_post(HitPre{b, disc}, ta.eps);
                    if (t < tmin) { tmin = t; idx = (int)sphere; }
                }
            };
            for (uint32_t k = 0; k < n; k += 4) {
                const float4 a0 = tile[k], c0 = tile[k + 1], a1 = tile[k + 2], c1 = tile[k + 3];
                const HitPre2 h01 = intersect_pre2(a0, c0, s.ox, s.oy, s.oz, s.dx, s.dy, s.dz);
                const HitPre2 h23 = intersect_pre2(a1, c1, s.ox, s.oy, s.oz, s.dx, s.dy, s.dz);
                const float m = fmaxf(fmaxf(h01.disc.x, h01.disc.y), fmaxf(h23.disc.x, h23.disc.y)); // NaNs drop out
                if (__any(m >= 0.0f)) {
                    hit(h01.b.x, h01.disc.x, base + k);
                    hit(h01.b.y, h01.disc.y, base + k + 1);
                    hit(h23.b.x, h23.disc.x, base + k + 2);
                    hit(h23.b.y, h23.disc.y, base + k + 3);
                }
            }
        }
        const uint32_t g = (idx < 0) ? ns - 1 : (uint32_t)idx;
        PathState n = s;
        shade_and_reflect<MODE>(n, tmin, cx[g], cy[g], cz[g], colx[g], coly[g], colz[g], idx == ta.light);
        if (ta.rr_start && d + 1 >= ta.rr_start) russian_roulette(n, rr_key, d);
        if (!fin) { s = n; ++traced; }
    }
    return traced;
}

// ---- trace: any scene through the host-built grid (apt_render_params.accel) ---------------------
// Per lane: the always-tested large spheres, then a 3D-DDA over the cells of the small ones.  Every
// candidate goes through the reference's exact arithmetic (intersect_pre/intersect_post), so the set
// of (t, sphere) pairs that can win is a subset of what the brute-force loop sees, and the traversal
// only drops spheres that cannot be hit: a sphere's box was inflated by `margin` when it was binned,
// the walk stops only once the nearest accepted root lies clearly before the exit of the current cell,
// and the arg-min is order independent (equal t -> lower sphere index, the brute-force loop's rule).
// The geometric argument needs a unit-length direction (the reference's roots are only the geometric
// ray parameters then): lanes whose |d|^2 is not within 1e-3 of 1, or not finite, test every sphere.
template <int MODE, bool RETIRE>
__device__ __forceinline__ uint32_t trace_grid(const float *__restrict__ sph, const uint32_t *__restrict__ grid,
                                               PathState &s, bool valid, const TraceArgs &ta, uint64_t path) {
    const GridHeader &h = *reinterpret_cast<const GridHeader *>(grid);
    const uint32_t ns = ta.ns;
    const uint32_t *large = grid + h.off_large, *cells = grid + h.off_cells, *items = grid + h.off_items;
    const float4 *geom = reinterpret_cast<const float4 *>(grid + h.off_geom);
    const float4 *item_geom = reinterpret_cast<const float4 *>(grid + h.off_item_geom);
    const float *colx = sph + 7 * (size_t)ns, *coly = sph + 8 * (size_t)ns, *colz = sph + 9 * (size_t)ns;
    const uint64_t rr_key = ta.rr_start ? rr_path_key(ta.seed, path) : 0;
    const int n0 = (int)h.n[0], n1 = (int)h.n[1], n2 = (int)h.n[2];
    uint32_t traced = 0, n_cells = 0, n_tests = 0; // statistics
    for (uint32_t d = 0; d < ta.depth; ++d) {
        const bool fin = !valid || (RETIRE && path_finished(s));
        if (RETIRE && __all(fin)) break;
        float tmin = kMissT;
        int idx = (MODE == kModeOracle) ? -1 : 0;
        auto test_geom = [&](const float4 g, uint32_t k) {
            ++n_tests;
            const HitPre hp = intersect_pre(g.x, g.y, g.z, g.w, s.ox, s.oy, s.oz, s.dx, s.dy, s.dz);
            if (hp.disc >= 0.0f) {
                const float t = intersect_post(hp, ta.eps);
                if (t < tmin || (t == tmin && (int)k < idx)) { tmin = t; idx = (int)k; }
            }
        };
        auto test = [&](uint32_t k) { test_geom(geom[k], k); };
        for (uint32_t i = 0; i < h.nlarge; ++i) test(large[i]); // wave-uniform: scalar loads
        const float dd = s.dx * s.dx + s.dy * s.dy + s.dz * s.dz;
        const bool unit = fabsf(dd - 1.0f) <= 1e-3f; // false for NaN/inf
        if (!fin && !unit) {
            for (uint32_t k = 0; k < ns; ++k) test(k);
        } else if (!fin) {
            // slab test against the grid box; all DDA state in scalars (no indexed arrays -> no scratch)
            float tn = 0.0f, tf = 3.0e38f;
            bool inbox = true;
            auto slab = [&](float o, float dv, float lo, float hi) {
                if (fabsf(dv) > 1e-20f) {
                    const float inv = 1.0f / dv, t1 = (lo - o) * inv, t2 = (hi - o) * inv;
                    tn = fmaxf(tn, fminf(t1, t2));
                    tf = fminf(tf, fmaxf(t1, t2));
                } else if (!(o >= lo && o <= hi)) inbox = false;
            };
            slab(s.ox, s.dx, h.gmin[0], h.gmax[0]);
            slab(s.oy, s.dy, h.gmin[1], h.gmax[1]);
            slab(s.oz, s.dz, h.gmin[2], h.gmax[2]);
            if (inbox && tn <= tf) {
                auto axis = [&](float o, float dv, float lo, float cellw, float invw, int na, int &c, int &step, float &tmax,
                                float &tdel) {
                    int ci = (int)floorf((o + dv * tn - lo) * invw);
                    ci = ci < 0 ? 0 : (ci >= na ? na - 1 : ci);
                    c = ci;
                    if (dv > 1e-20f) { step = 1; tmax = (lo + (float)(ci + 1) * cellw - o) / dv; tdel = cellw / dv; }
                    else if (dv < -1e-20f) { step = -1; tmax = (lo + (float)ci * cellw - o) / dv; tdel = -cellw / dv; }
                    else { step = 0; tmax = 3.0e38f; tdel = 3.0e38f; }
                };
                int c0, c1, c2, st0, st1, st2;
                float tm0, tm1, tm2, td0, td1, td2;
                axis(s.ox, s.dx, h.gmin[0], h.cell[0], h.inv_cell[0], n0, c0, st0, tm0, td0);
                axis(s.oy, s.dy, h.gmin[1], h.cell[1], h.inv_cell[1], n1, c1, st1, tm1, td1);
                axis(s.oz, s.dz, h.gmin[2], h.cell[2], h.inv_cell[2], n2, c2, st2, tm2, td2);
                const int max_steps = n0 + n1 + n2 + 3;
                for (int it = 0; it < max_steps; ++it) {
                    const uint32_t cell = (uint32_t)((c2 * n1 + c1) * n0 + c0);
                    const uint32_t b = cells[cell], e = cells[cell + 1];
                    ++n_cells;
                    uint32_t i = b;
                    for (; i + 2 <= e; i += 2) { // two candidates per step: four independent loads in flight
                        const float4 ga = item_geom[i], gb = item_geom[i + 1];
                        const uint32_t ka = items[i], kb = items[i + 1];
                        test_geom(ga, ka);
                        test_geom(gb, kb);
                    }
                    if (i < e) test_geom(item_geom[i], items[i]);
                    const float te = fminf(tm0, fminf(tm1, tm2));                      // parameter at which the ray leaves this cell
                    if (tmin < te - (1e-3f * fabsf(te) + h.margin)) break;             // nothing nearer can lie ahead
                    if (tm0 <= tm1 && tm0 <= tm2) { c0 += st0; tm0 += td0; if ((unsigned)c0 >= (unsigned)n0) break; }
                    else if (tm1 <= tm2) { c1 += st1; tm1 += td1; if ((unsigned)c1 >= (unsigned)n1) break; }
                    else { c2 += st2; tm2 += td2; if ((unsigned)c2 >= (unsigned)n2) break; }
                }
            }
        }
        if (!fin) { // per-lane code anyway: shade in place (no second copy of the path state in registers)
            const uint32_t g = (idx < 0) ? ns - 1 : (uint32_t)idx;
            const float4 gc = geom[g];
            shade_and_reflect<MODE>(s, tmin, gc.x, gc.y, gc.z, colx[g], coly[g], colz[g], idx == ta.light);
            if (ta.rr_start && d + 1 >= ta.rr_start) russian_roulette(s, rr_key, d);
            ++traced;
        }
    }
    if (ta.traced) { // statistics: cells visited / candidates tested (per lane, summed over the wave)
        unsigned long long c = n_cells, t = n_tests;
        for (int off = 32; off > 0; off >>= 1) { c += __shfl_xor(c, off, 64); t += __shfl_xor(t, off, 64); }
        if ((threadIdx.x & 63) == 0) { atomicAdd(ta.traced + 1, c); atomicAdd(ta.traced + 2, t); }
    }
    return traced;
}

// spheres.bin layout [10][8]: r2, x, y, z, em*3, col*3 (gen_data.py:106-127, rt_helper.h:93-102)
__device__ __forceinline__ void load_scene8(const float *__restrict__ sph, Scene8 &sc, float4 *tab) {
#pragma unroll
    for (int k = 0; k < 8; ++k) { // constant offsets from a uniform read-only pointer: scalar loads
        sc.r2[k] = sph[k]; sc.cx[k] = sph[8 + k]; sc.cy[k] = sph[16 + k]; sc.cz[k] = sph[24 + k];
    }
    if (threadIdx.x < 8) {
        const int k = threadIdx.x;
        tab[2 * k] = make_float4(sph[8 + k], sph[16 + k], sph[24 + k], sph[k]);
        tab[2 * k + 1] = make_float4(sph[56 + k], sph[64 + k], sph[72 + k], 0.0f);
    }
    __syncthreads();
}

// render.cpp:194-196 multiplies by the literal 12; with APT_FLAG_EMISSION the light's emission planes
// (spheres.bin rows 4..6, never read by the reference) are used instead: identical for the reference
// scene, whose light emits (12,12,12).  Wave-uniform scalar loads.
struct Gain3 { float r, g, b; };
__device__ __forceinline__ Gain3 load_gain(const float *__restrict__ sph, const TraceArgs &ta) {
    if (ta.emission) {
        const size_t ns = ta.ns, l = (size_t)ta.light;
        return Gain3{sph[4 * ns + l], sph[5 * ns + l], sph[6 * ns + l]};
    }
    return Gain3{ta.gain, ta.gain, ta.gain};
}

__device__ __forceinline__ void count_traced(const TraceArgs &ta, uint32_t traced) {
    if (ta.traced) { // one atomic per wave
        unsigned long long t = traced;
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
        if ((threadIdx.x & 63) == 0 && t) atomicAdd(ta.traced, t);
    }
}

// ---- kernel: rays from a buffer ---------------------------------------------------------
template <int MODE, int SC, bool RETIRE>
__global__ __launch_bounds__(kBlock, SC == kSceneGrid ? APT_GRID_WAVES : 1) void render_paths_kernel(const float *__restrict__ rays,
                                                              const float *__restrict__ sph,
                                                              float *__restrict__ colors, uint64_t n_total,
                                                              uint64_t begin, uint64_t count, TraceArgs ta) {
    constexpr bool NS8 = SC == kScene8;
    __shared__ float4 tab[16];
    __shared__ float4 tile[SC == kSceneTiles ? kTile : 1];
    Scene8 sc;
    if (NS8) load_scene8(sph, sc, tab);
    const uint64_t local = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool valid = local < count;
    const uint64_t p = begin + (valid ? local : 0);
    PathState s;                                          // CopyIn: render.cpp:82-101
    path_init(s, rays[p], rays[n_total + p], rays[2 * n_total + p], rays[3 * n_total + p], rays[4 * n_total + p],
              rays[5 * n_total + p]);
    uint32_t traced;
    if (SC == kScene8) traced = trace_ns8<MODE, RETIRE>(sc, tab, s, valid, ta, p);
    else if (SC == kSceneGrid) traced = trace_grid<MODE, RETIRE>(sph, ta.grid, s, valid, ta, p);
    else traced = trace_dyn<MODE, RETIRE>(sph, tile, s, valid, ta, p);
    if (valid) {                                          // render.cpp:194-196, CopyOut :210-223
        const Gain3 gain = load_gain(sph, ta);
        colors[p] = s.rx * gain.r;
        colors[n_total + p] = s.ry * gain.g;
        colors[2 * n_total + p] = s.rz * gain.b;
    }
    count_traced(ta, valid ? traced : 0);
}

// ---- kernel: rays from a buffer, with active-ray compaction (APT_FLAG_RETIRE, Ns == 8) ------------
// Buffer mode has no ordering constraint on its outputs (colour p is stored to colors[p]), so the
// wave-level queue is simple: every wave owns kQueueChunk consecutive paths; a lane whose path is
// finished (alive bit cleared, throughput zero, depth reached) takes the next unissued path of the
// chunk -- ballot of the idle lanes, mbcnt prefix rank, p = next + rank -- and loads its ray.
constexpr uint32_t kQueueChunk = 64 * 16;
template <int MODE>
__global__ __launch_bounds__(kBlock) void render_paths_queue_kernel(const float *__restrict__ rays,
                                                                    const float *__restrict__ sph,
                                                                    float *__restrict__ colors, uint64_t n_total,
                                                                    uint64_t begin, uint64_t count, TraceArgs ta) {
    __shared__ float4 tab[16];
    Scene8 sc;
    load_scene8(sph, sc, tab);
    const Gain3 gain = load_gain(sph, ta);
    const uint64_t wave = (uint64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    uint64_t next = wave * kQueueChunk;                         // wave-uniform
    const uint64_t end = min(count, next + kQueueChunk);
    uint32_t depth_left = 0, traced = 0;
    uint64_t cur = 0, cur_key = 0;
    PathState s;
    path_init(s, 0.f, 0.f, 0.f, 0.f, 0.f, 1.f);
    for (;;) {
        const bool want = depth_left == 0;
        const unsigned long long wants = __ballot(want);
        if (next < end && wants) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(wants >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)wants, 0u));
            const uint64_t remaining = end - next;
            if (want && rank < remaining) {
                cur = begin + next + rank;
                path_init(s, rays[cur], rays[n_total + cur], rays[2 * n_total + cur], rays[3 * n_total + cur],
                          rays[4 * n_total + cur], rays[5 * n_total + cur]);
                depth_left = ta.depth;
                if (ta.rr_start) cur_key = rr_path_key(ta.seed, cur);
                if (ta.depth == 0) { colors[cur] = gain.r; colors[n_total + cur] = gain.g; colors[2 * n_total + cur] = gain.b; }
            }
            next += min((uint64_t)__popcll(wants), remaining);
        }
        const bool active = depth_left != 0;
        if (!__any(active)) {
            if (next >= end) break;
            continue;
        }
        PathState nx;
        bool redo = bounce_ns8<MODE, true>(sc, tab, s, nx, ta);
        redo = redo && active;
        if (__builtin_expect(__any(redo), 0)) { // exact re-run, see trace_ns8
            asm volatile("" ::: "memory");
            (void)bounce_ns8<MODE, false>(sc, tab, s, nx, ta);
        }
        if (ta.rr_start && ta.depth - depth_left + 1 >= ta.rr_start) russian_roulette(nx, cur_key, ta.depth - depth_left);
        s = nx;
        traced += active ? 1u : 0u;
        depth_left -= active ? 1u : 0u;
        if (active && (depth_left == 0 || path_finished(s))) {
            depth_left = 0;
            colors[cur] = s.rx * gain.r;
            colors[n_total + cur] = s.ry * gain.g;
            colors[2 * n_total + cur] = s.rz * gain.b;
        }
    }
    count_traced(ta, traced);
}

// ---- kernel: fused frame ----------------------------------------------------------------
struct FrameArgs {
    Camera cam;
    uint32_t width, height, samples;
    uint64_t seed;
    uint64_t pixel_begin, pixel_count;
    float *fb;       // [3][pixel_count]
    uint8_t *fb_u8;  // [pixel_count][3] or null
};

// GROUP lanes share one sub-pixel: lane j of the group owns numpy's pairwise accumulator
// r[j] (samples j, 8+j, 16+j, ...), so the summation order of np.mean is reproduced with
// a 3-step butterfly and no shared memory.  GROUP == 1 serves samples < 8 (numpy sums
// those sequentially).
template <int MODE, int SC, int GROUP, bool RETIRE>
__global__ __launch_bounds__(kBlock, (RETIRE && SC == kScene8 && GROUP == 8) ? 5 : (SC == kSceneGrid ? APT_GRID_WAVES : APT_FULL_WAVES)) void render_frame_kernel(const float *__restrict__ sph, FrameArgs fa,
                                                              TraceArgs ta, LeafProg lp) {
    constexpr bool NS8 = SC == kScene8;
    __shared__ float4 tab[16];
    __shared__ float4 tile[SC == kSceneTiles ? kTile : 1];
    extern __shared__ float dyn_lds[];
    float *stack_lds = dyn_lds;                                            // [kMaxStack][3][kStackSlots] when lp.nleaves > 1
    float *queue_lds = dyn_lds + (lp.nleaves > 1 ? kMaxStack * 3 * kStackSlots : 0); // [waves][3][8*maxleaf] (refill)
    // The camera frame (14 doubles) is only needed by ray-generate; parked in LDS it does not
    // occupy 28 SGPRs across the bounce loop (they spilled to VGPR lanes otherwise).
    __shared__ Camera cam;
    if (threadIdx.x < sizeof(Camera) / sizeof(double)) (&cam.pos[0])[threadIdx.x] = (&fa.cam.pos[0])[threadIdx.x];
    Scene8 sc;
    if (NS8) load_scene8(sph, sc, tab);
    else __syncthreads();

    const uint32_t lane = threadIdx.x & 63;
    const uint64_t L = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint32_t j = (GROUP == 8) ? (uint32_t)(L & 7) : 0u;
    const uint32_t sub = (uint32_t)(L / GROUP) & 3u;
    const uint64_t pl = L / (4 * GROUP);
    const bool valid = pl < fa.pixel_count;
    const uint64_t q = fa.pixel_begin + (valid ? pl : 0);
    const uint32_t pi = (uint32_t)(q / fa.height), pj = (uint32_t)(q % fa.height);
    const uint32_t sy = sub >> 1, sx = sub & 1;
    const uint64_t pbase = (q * 4 + sub) * fa.samples;
    uint32_t traced = 0;

    const Gain3 gain = load_gain(sph, ta);
    struct Col { float r, g, b; };
    auto sample = [&](uint32_t k) -> Col {
        double u1, u2;
        path_uniforms(fa.seed, pbase + k, u1, u2);
        float rox, roy, roz, rdx, rdy, rdz;
        camera_ray(cam, fa.width, fa.height, pi, pj, sy, sx, u1, u2, rox, roy, roz, rdx, rdy, rdz);
        PathState s;
        path_init(s, rox, roy, roz, rdx, rdy, rdz);
        if (SC == kScene8) traced += trace_ns8<MODE, RETIRE>(sc, tab, s, valid, ta, pbase + k);
        else if (SC == kSceneGrid) traced += trace_grid<MODE, RETIRE>(sph, ta.grid, s, valid, ta, pbase + k);
        else traced += trace_dyn<MODE, RETIRE>(sph, tile, s, valid, ta, pbase + k);
        return Col{s.rx * gain.r, s.ry * gain.g, s.rz * gain.b};
    };
    auto add = [](const Col &a, const Col &b) { return Col{a.r + b.r, a.g + b.g, a.b + b.b}; };

    float res[3] = {0.0f, 0.0f, 0.0f};
    uint32_t start = 0;
    int sp = 0;
    for (uint32_t leaf = 0; leaf < lp.nleaves; ++leaf) {
        const uint32_t n = lp.len(leaf);
        float acc[3];
        if (GROUP == 1) { // n < 8: res = 0; res += a[i]
            Col a = {0.0f, 0.0f, 0.0f};
            for (uint32_t k = 0; k < n; ++k) a = add(a, sample(start + k));
            acc[0] = a.r; acc[1] = a.g; acc[2] = a.b;
        } else {          // 8 <= n <= 128: r[j] chains, tree, tail
            const uint32_t nfull = n & ~7u;
            if (RETIRE && NS8) {
                // Active-ray compaction with a wave-level work queue.  The 8 sub-pixel groups of the
                // wave have 8*nfull samples in this leaf; instead of binding sample k of group g to
                // lane (g, k mod 8), any lane that runs out of work takes the next unissued sample:
                // a ballot of the lanes with an empty one-ray slot, a prefix count (mbcnt) as the
                // rank inside the batch, item = next + rank.  Finished colours are parked in a
                // per-wave LDS array indexed by the sample, and lane (g, j) then adds its own chain
                // j, 8+j, ... from there IN ORDER, so numpy's summation order is untouched and the
                // frame stays bit-identical.  Ray-generate (float64, the expensive part) runs for
                // the whole wave only when >= kRefillLanes lanes want a ray or nothing else is left.
                float *colq = queue_lds + (size_t)(threadIdx.x >> 6) * 3u * 8u * lp.maxleaf; // [3][8*maxleaf]
                const uint32_t total = 8u * nfull;        // items of this wave in this leaf (uniform)
                const uint32_t qstride = 8u * lp.maxleaf;
                uint32_t next = 0;                        // first unissued item (uniform)
                uint32_t depth_left = 0, cur_item = 0, slot_item = 0;
                uint64_t cur_key = 0, slot_key = 0;       // Russian-roulette keys of the running / waiting path
                uint32_t n_bounce_exec = 0, n_gen_exec = 0; // wave-level executions (statistics only)
                float sl_ox = 0.f, sl_oy = 0.f, sl_oz = 0.f, sl_dx = 0.f, sl_dy = 0.f, sl_dz = 1.f; // the one-ray slot
                bool slot_full = false, slot_valid = false, cur_valid = false;
                PathState s;
                path_init(s, 0.f, 0.f, 0.f, 0.f, 0.f, 1.f);
                for (;;) {
                    const bool want = !slot_full;
                    const unsigned long long wants = __ballot(want);
                    const bool busy_any = __any(depth_left != 0 || slot_full);
                    if (next < total && wants && ((uint32_t)__popcll(wants) >= ta.refill_lanes || !busy_any)) {
                        ++n_gen_exec;
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(wants >> 32),
                                              __builtin_amdgcn_mbcnt_lo((uint32_t)wants, 0u));
                        const uint32_t remaining = total - next;
                        const bool take = want && rank < remaining;
                        // group coordinates come from the first lane of the item's group; every lane
                        // of the wave takes part in the shuffles (a masked-off source lane would
                        // return garbage), lanes that do not take an item use item 0
                        const uint32_t item = take ? next + rank : 0u;
                        const uint32_t g = item / nfull, k = item - g * nfull;
                        const int src = (int)(8u * g);
                        const uint32_t gpi = __shfl(pi, src, 64), gpj = __shfl(pj, src, 64);
                        const uint32_t gsub = __shfl(sub, src, 64);
                        const uint32_t blo = __shfl((uint32_t)pbase, src, 64), bhi = __shfl((uint32_t)(pbase >> 32), src, 64);
                        const bool gvalid = __shfl((int)valid, src, 64) != 0;
                        if (take) {
                            slot_valid = gvalid;
                            double u1, u2;
                            const uint64_t path = (((uint64_t)bhi << 32) | blo) + start + k;
                            if (ta.rr_start) slot_key = rr_path_key(ta.seed, path);
                            path_uniforms(fa.seed, path, u1, u2);
                            camera_ray(cam, fa.width, fa.height, gpi, gpj, gsub >> 1, gsub & 1u, u1, u2, sl_ox, sl_oy, sl_oz, sl_dx,
                                       sl_dy, sl_dz);
                            slot_item = item;
                            slot_full = true;
                        }
                        next += min((uint32_t)__popcll(wants), remaining);
                    }
                    if (depth_left == 0 && slot_full) { // start the waiting ray
                        path_init(s, sl_ox, sl_oy, sl_oz, sl_dx, sl_dy, sl_dz);
                        cur_item = slot_item;
                        cur_key = slot_key;
                        cur_valid = slot_valid;
                        depth_left = ta.depth;
                        slot_full = false;
                        if (ta.depth == 0 || !cur_valid) { // depth 0, or a group past the image: colour = gain
                            depth_left = 0;
                            colq[cur_item] = gain.r; colq[qstride + cur_item] = gain.g; colq[2 * qstride + cur_item] = gain.b;
                        }
                    }
                    const bool active = depth_left != 0;
                    if (!__any(active)) {
                        if (next >= total && !__any(slot_full)) break;
                        continue;
                    }
                    ++n_bounce_exec;
                    PathState nx;
                    bool redo = bounce_ns8<MODE, true>(sc, tab, s, nx, ta);
                    redo = redo && active;
                    if (__builtin_expect(__any(redo), 0)) { // exact re-run, see trace_ns8
                        asm volatile("" ::: "memory");
                        (void)bounce_ns8<MODE, false>(sc, tab, s, nx, ta);
                    }
                    if (ta.rr_start && ta.depth - depth_left + 1 >= ta.rr_start) // 0-based bounce index = depth - depth_left
                        russian_roulette(nx, cur_key, ta.depth - depth_left);
                    // inactive lanes computed on stale state; whatever they hold is overwritten when
                    // they start their next ray, so the update itself needs no mask
                    s = nx;
                    traced += active ? 1u : 0u;
                    depth_left -= active ? 1u : 0u;
                    if (active && (depth_left == 0 || path_finished(s))) {
                        depth_left = 0;
                        colq[cur_item] = s.rx * gain.r;
                        colq[qstride + cur_item] = s.ry * gain.g;
                        colq[2 * qstride + cur_item] = s.rz * gain.b;
                    }
                }
                __syncthreads(); // colours of the whole leaf are in LDS (only wave-local data is read back)
                {   // lane (g, j) adds samples j, 8+j, ... of its own group, in order: numpy's r[j] chain
                    const uint32_t base = (lane >> 3) * nfull + j;
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) acc[ch] = colq[ch * qstride + base];
                    for (uint32_t i8 = 8; i8 < nfull; i8 += 8) {
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) acc[ch] = acc[ch] + colq[ch * qstride + base + i8];
                    }
                }
                __syncthreads(); // before the next leaf reuses the array
                if (ta.traced && lane == 0) { // lane-slots spent: executions x 64
                    atomicAdd(ta.traced + 1, 64ull * n_bounce_exec);
                    atomicAdd(ta.traced + 2, 64ull * n_gen_exec);
                }
            } else {
                Col a = sample(start + j);
                for (uint32_t i8 = 8; i8 < nfull; i8 += 8) a = add(a, sample(start + i8 + j));
                acc[0] = a.r; acc[1] = a.g; acc[2] = a.b;
            }
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) { // ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7))
                float v = acc[ch];
                v = v + __shfl_xor(v, 1, 64);
                v = v + __shfl_xor(v, 2, 64);
                v = v + __shfl_xor(v, 4, 64);
                acc[ch] = v;
            }
            const uint32_t nt = n - nfull;
            if (nt) { // res += a[i] for the n % 8 trailing samples, in order
                const Col c = sample(start + nfull + (j < nt ? j : 0));
                for (uint32_t t = 0; t < nt; ++t) {
                    const int src = (int)((lane & ~7u) + t);
                    acc[0] = acc[0] + __shfl(c.r, src, 64);
                    acc[1] = acc[1] + __shfl(c.g, src, 64);
                    acc[2] = acc[2] + __shfl(c.b, src, 64);
                }
            }
        }
        start += n;
        if (lp.nleaves == 1) {
            res[0] = acc[0]; res[1] = acc[1]; res[2] = acc[2];
        } else { // pairwise(left) + pairwise(right), innermost first
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) stack_lds[(sp * 3 + ch) * kStackSlots + (threadIdx.x >> 3)] = acc[ch];
            ++sp;
            for (uint32_t m = 0; m < lp.ncomb(leaf); ++m) {
                --sp;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const float a = stack_lds[((sp - 1) * 3 + ch) * kStackSlots + (threadIdx.x >> 3)];
                    const float b = stack_lds[(sp * 3 + ch) * kStackSlots + (threadIdx.x >> 3)];
                    stack_lds[((sp - 1) * 3 + ch) * kStackSlots + (threadIdx.x >> 3)] = a + b;
                }
            }
        }
    }
    if (lp.nleaves > 1) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) res[ch] = stack_lds[ch * kStackSlots + (threadIdx.x >> 3)];
    }

    // decode_color: data_visualization.py:36-57
    const float fs = (float)fa.samples;
    const int gbase = (int)(lane & ~(uint32_t)(4 * GROUP - 1));
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const float mean = res[ch] / fs;            // np.mean: float32 sum / count
        double acc = 0.0;                           // :38 sum_color = zeros (float64)
#pragma unroll
        for (int sq = 0; sq < 4; ++sq) acc = acc + (double)__shfl(mean, gbase + sq * GROUP, 64); // :41-45
        const double v = acc / 4;                   // :46
        const double cl = v < 0 ? 0 : (v > 1 ? 1 : v); // :54
        if (valid && (lane & (4 * GROUP - 1)) == 0) {
            fa.fb[(uint64_t)ch * fa.pixel_count + pl] = (float)cl;
            if (fa.fb_u8) fa.fb_u8[pl * 3 + ch] = (uint8_t)(cl * 255); // :55-57 truncation
        }
    }
    count_traced(ta, valid ? traced : 0);
}

// ---- kernel: first-hit debug oracle (gen_data.py:134-188 test_scene) ---------------------------
// out[3][N]: emission of the light when it is the first hit, the sphere's colour otherwise, 0 when
// nothing is hit.  One lane per ray, spheres read straight from the [10][Ns] planes (L2-resident).
__global__ __launch_bounds__(kBlock) void test_scene_kernel(const float *__restrict__ rays,
                                                            const float *__restrict__ sph, float *__restrict__ out,
                                                            uint64_t n_total, uint32_t ns, int32_t light, float eps) {
    const uint64_t p = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= n_total) return;
    const float ox = rays[p], oy = rays[n_total + p], oz = rays[2 * n_total + p];
    const float dx = rays[3 * n_total + p], dy = rays[4 * n_total + p], dz = rays[5 * n_total + p];
    float mind = kMissT;
    int id = -1;
    for (uint32_t k = 0; k < ns; ++k)
        test_scene_sphere(sph[ns + k], sph[2 * (size_t)ns + k], sph[3 * (size_t)ns + k], sph[k], ox, oy, oz, dx, dy, dz,
                          eps, (int)k, mind, id);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = 0.0f;
        if (id >= 0) v = (id == light) ? sph[(size_t)(4 + c) * ns + id] : sph[(size_t)(7 + c) * ns + id]; // :175-180
        out[(uint64_t)c * n_total + p] = v;
    }
}

// ---- kernel: device gen_rays (counter RNG) -----------------------------------------------
__global__ __launch_bounds__(kBlock) void gen_rays_kernel(Camera cam, uint32_t width, uint32_t height,
                                                          uint32_t samples, uint64_t seed, uint64_t n_total,
                                                          uint64_t begin, uint64_t count, float *__restrict__ rays) {
    const uint64_t local = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (local >= count) return;
    const uint64_t p = begin + local;
    uint32_t i, j, sy, sx;
    path_coords(p, height, samples, i, j, sy, sx);
    double u1, u2;
    path_uniforms(seed, p, u1, u2);
    float rox, roy, roz, rdx, rdy, rdz;
    camera_ray(cam, width, height, i, j, sy, sx, u1, u2, rox, roy, roz, rdx, rdy, rdz);
    rays[p] = rox; rays[n_total + p] = roy; rays[2 * n_total + p] = roz;
    rays[3 * n_total + p] = rdx; rays[4 * n_total + p] = rdy; rays[5 * n_total + p] = rdz;
}

// ---- kernel: device gen_rays, bit-exact with the reference's MT19937 stream -------------------
// np.random.rand() takes two MT19937 words per double and gen_rays two doubles per path, in path
// order (gen_data.py:32-40), so output block b of the generator (624 words) is exactly paths
// [156b, 156b+156).  One workgroup per checkpoint: load the raw state of block cb = i*stride into
// LDS, emit that block, then `twist` forward block by block.  The twist is the textbook 3-phase
// parallel form: x[i] depends on x[i], x[i+1] and x[i+397], so [0,227), [227,454), [454,624) can each
// be updated at once (read, barrier, write, barrier).
constexpr int kMtN = 624, kMtM = 397, kPathsPerBlock = 156;

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

__global__ __launch_bounds__(kBlock) void gen_rays_mt_kernel(const uint32_t *__restrict__ checkpoints, uint32_t stride,
                                                             uint64_t num_blocks, Camera cam, uint32_t width,
                                                             uint32_t height, uint32_t samples, uint64_t n_total,
                                                             uint64_t begin, uint64_t end, float *__restrict__ rays) {
    __shared__ uint32_t mt[kMtN];
    const uint64_t cb = (uint64_t)blockIdx.x * stride;         // first output block of this workgroup
    for (int i = threadIdx.x; i < kMtN; i += kBlock) mt[i] = checkpoints[(uint64_t)blockIdx.x * kMtN + i];
    __syncthreads();
    const uint64_t last = min(cb + stride, num_blocks);
    for (uint64_t blk = cb; blk < last; ++blk) {
        if (blk != cb) { // twist to the next block
            const int t = threadIdx.x;
            const int lo[3] = {0, 227, 454}, hi[3] = {227, 454, 624};
#pragma unroll
            for (int ph = 0; ph < 3; ++ph) {
                const int i = lo[ph] + t;
                uint32_t v = 0;
                const bool on = i < hi[ph];
                if (on) {
                    const uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % kMtN] & 0x7fffffffu);
                    v = mt[(i + kMtM) % kMtN] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
                }
                __syncthreads();
                if (on) mt[i] = v;
                __syncthreads();
            }
        }
        const uint64_t p = blk * kPathsPerBlock + threadIdx.x;
        if (threadIdx.x < kPathsPerBlock && p >= begin && p < end) {
            const uint32_t a1 = mt_temper(mt[4 * threadIdx.x]) >> 5, b1 = mt_temper(mt[4 * threadIdx.x + 1]) >> 6;
            const uint32_t a2 = mt_temper(mt[4 * threadIdx.x + 2]) >> 5, b2 = mt_temper(mt[4 * threadIdx.x + 3]) >> 6;
            const double u1 = ((double)a1 * 67108864.0 + (double)b1) / 9007199254740992.0; // random_sample
            const double u2 = ((double)a2 * 67108864.0 + (double)b2) / 9007199254740992.0;
            uint32_t i, j, sy, sx;
            path_coords(p, height, samples, i, j, sy, sx);
            float rox, roy, roz, rdx, rdy, rdz;
            camera_ray(cam, width, height, i, j, sy, sx, u1, u2, rox, roy, roz, rdx, rdy, rdz);
            rays[p] = rox; rays[n_total + p] = roy; rays[2 * n_total + p] = roz;
            rays[3 * n_total + p] = rdx; rays[4 * n_total + p] = rdy; rays[5 * n_total + p] = rdz;
        }
    }
}

// ---- kernel: device decode_color -----------------------------------------------------------
__device__ float pairwise_leaf(const float *a, uint32_t n) { // numpy pairwise_sum, n <= 128
    if (n < 8) {
        float r = 0.0f;
        for (uint32_t i = 0; i < n; ++i) r = r + a[i];
        return r;
    }
    float r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = a[k];
    uint32_t i;
    for (i = 8; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = r[k] + a[i + k];
    }
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res = res + a[i];
    return res;
}

// one thread per (pixel, channel, sub-pixel); 4 adjacent lanes combine in float64
__global__ __launch_bounds__(kBlock) void decode_color_kernel(const float *__restrict__ colors, uint32_t samples,
                                                              uint64_t npix, LeafProg lp, float *__restrict__ fb,
                                                              uint8_t *__restrict__ fb_u8) {
    const uint64_t L = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint32_t sub = (uint32_t)(L & 3);
    const uint64_t pc = L >> 2; // pixel * 3 + channel, channel-major: pc = ch * npix + q
    const bool valid = pc < 3 * npix;
    const uint64_t ch = valid ? pc / npix : 0, q = valid ? pc % npix : 0;
    const uint64_t n_total = npix * 4 * samples;
    const float *a = colors + ch * n_total + (q * 4 + sub) * samples;
    float st[kMaxStack];
    int sp = 0;
    uint32_t start = 0;
    for (uint32_t leaf = 0; leaf < lp.nleaves; ++leaf) {
        st[sp++] = pairwise_leaf(a + start, lp.len(leaf));
        start += lp.len(leaf);
        for (uint32_t m = 0; m < lp.ncomb(leaf); ++m) { --sp; st[sp - 1] = st[sp - 1] + st[sp]; }
    }
    const float mean = st[0] / (float)samples;
    const int gbase = (int)((threadIdx.x & 63) & ~3u);
    double acc = 0.0;
#pragma unroll
    for (int sq = 0; sq < 4; ++sq) acc = acc + (double)__shfl(mean, gbase + sq, 64);
    const double v = acc / 4;
    const double cl = v < 0 ? 0 : (v > 1 ? 1 : v);
    if (valid && sub == 0) {
        fb[ch * npix + q] = (float)cl;
        if (fb_u8) fb_u8[q * 3 + ch] = (uint8_t)(cl * 255);
    }
}

// ---- kernel: exhaustive self-test of the fast correctly-rounded sqrt ------------------------
// Every float bit pattern in [begin, begin+count): variant(x) must equal sqrtf(x) bit for bit
// (any NaN == any NaN) unless the variant asks for the fallback (|x| < 2^-96), in which case the
// hot loop would have used sqrtf() anyway.  Counts mismatches; remembers the first one.
__global__ __launch_bounds__(kBlock) void selftest_sqrt_kernel(int variant, uint64_t begin, uint64_t count,
                                                               unsigned long long *result) {
#if defined(__HIP_DEVICE_COMPILE__) // the sqrt variants are device-only builtins
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    unsigned long long bad = 0, first = ~0ull;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += stride) {
        const uint32_t bits = (uint32_t)(begin + i);
        const float x = __uint_as_float(bits);
        float amin = 1.0f;
        const float got = variant == 0 ? sqrt_rn_core(x, amin) : variant == 1 ? sqrt_rn_markstein(x, amin)
                        : variant == 2 ? sqrt_rn_rsq1(x, amin) : sqrt_rn_rsq2(x, amin);
        const float want = sqrtf(x);
        const bool fallback = amin < 0x1p-96f;
        const bool same = (__float_as_uint(got) == __float_as_uint(want)) || (got != got && want != want);
        if (!same && !fallback) { ++bad; if (first == ~0ull) first = bits; }
    }
    if (bad) {
        atomicAdd(&result[0], bad);
        atomicMin(&result[1], first);
    }
#endif
}

// ---- kernel: self-test of the shared-reciprocal divide --------------------------------------
// Operand set i of [begin, begin+count): three numerators built from a counter hash, the divisor
// formed from them exactly as the shading step does (sqrt of the sum of squares); a quarter of the
// sets use special mantissas (all ones, 1.0, powers of two, one-bit neighbours), exponents at the
// edges of the accepted range and signed zeros.  Every set the validity flags accept must give the
// three quotients of the plain `/` bit for bit.
__global__ __launch_bounds__(kBlock) void selftest_div3_kernel(uint64_t begin, uint64_t count,
                                                               unsigned long long *result) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    unsigned long long bad = 0, first = ~0ull, accepted = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += stride) {
        uint64_t h = splitmix64(begin + i);
        float v[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            h = splitmix64(h);
            // exponents over the whole accepted range and a little beyond it on both sides
            uint32_t man = (uint32_t)h & 0x7fffffu, ex = 127u - 100u + (uint32_t)((h >> 23) % 134u), sg = (uint32_t)(h >> 63);
            if (((begin + i) & 3u) == 0u) {
                const uint32_t pick = (uint32_t)(h >> 40) & 7u;
                man = pick == 0 ? 0x7fffffu : pick == 1 ? 0u : pick == 2 ? 1u : pick == 3 ? 0x7ffffeu
                    : pick == 4 ? 0x400000u : pick == 5 ? 0x3fffffu : pick == 6 ? 0x400001u : man;
                if (((h >> 44) & 3u) == 0u) ex = ((h >> 46) & 1u) ? 127u - 96u : 127u + 29u;
            }
            v[k] = __uint_as_float((sg << 31) | (ex << 23) | man);
        }
        if (((begin + i) & 63u) == 1u) v[((begin + i) >> 6) % 3u] = ((begin + i) & 64u) ? 0.0f : -0.0f; // zero numerators
        // the divisor exactly as the shading step forms it (K-mode order; O-mode differs by one rounding)
        float len2 = 0.0f + v[0] * v[0];
        len2 = len2 + v[1] * v[1];
        len2 = len2 + v[2] * v[2];
        const float d = sqrtf(len2);
        float ux, uy, uz, amin = 1.0f;
        uint32_t hiflag = 0;
        div3_shared(v[0], v[1], v[2], d, len2, ux, uy, uz, amin, hiflag);
        if (amin < 0x1p-96f || (int32_t)hiflag < 0) continue; // the kernel redoes these with '/'
        ++accepted;
        const float wx = v[0] / d, wy = v[1] / d, wz = v[2] / d;
        if (__float_as_uint(ux) != __float_as_uint(wx) || __float_as_uint(uy) != __float_as_uint(wy) ||
            __float_as_uint(uz) != __float_as_uint(wz)) { ++bad; if (first == ~0ull) first = begin + i; }
    }
    if (bad) { atomicAdd(&result[0], bad); atomicMin(&result[1], first); }
    atomicAdd(&result[2], accepted);
#endif
}

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
    const uint64_t lanes = npix * 3 * 4;
    const uint64_t blocks = (lanes + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return fail(APT_ERR_ARG, "image too large for one launch%s");
    hipLaunchKernelGGL(decode_color_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, colors,
                       p->samples, npix, lp, fb, fb_u8);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? APT_OK : hip_fail(e);
}

} // extern "C"
