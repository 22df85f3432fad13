// pt_trace.h -- device-side scene access and bounce loops of the render kernels (included by
// render_kernels.hip only).  Three ways a scene reaches the lanes:
//   trace_ns8   the reference's 8 spheres: geometry in SGPRs, packed sphere pairs, integer-key arg-min
//   trace_dyn   any Ns, brute force over LDS-staged tiles of sphere pairs
//   trace_grid  any Ns through the host-built uniform grid (per-lane DDA)
// All of them apply the reference's arithmetic from pt_core.h; see DESIGN.md section 4.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
            const HitPre hp = intersect_pre(g.x, g.y, g.z, g.w, s.ox, s.oy, s.oz, s.dx, s.dy, s.dz);
            if (hp.disc >= 0.0f) {
                const float t = intersect_post(hp, ta.eps);
                if (t < tmin || (t == tmin && (int)k < idx)) { tmin = t; idx = (int)k; }
            }
        };
        auto test = [&](uint32_t k) { ++n_tests; test_geom(geom[k], k); };
        // A candidate of the walk is identified by its position in the item list; the sphere index (one more
        // dependent load per candidate) is only fetched when it matters: on an exact tie of t -- mostly the same
        // sphere met again in the next cell -- and once at the end for the winner.
        uint32_t pos = ~0u; // item position of the running minimum, ~0u while `idx` itself is authoritative
        auto test_item = [&](const float4 g, uint32_t i) {
            const HitPre hp = intersect_pre(g.x, g.y, g.z, g.w, s.ox, s.oy, s.oz, s.dx, s.dy, s.dz);
            if (hp.disc >= 0.0f) {
                const float t = intersect_post(hp, ta.eps);
                if (t < tmin) { tmin = t; pos = i; }
                else if (t == tmin) {
                    const int cur = (pos != ~0u) ? (int)items[pos] : idx;
                    if ((int)items[i] < cur) pos = i;
                }
            }
        };
        // The large spheres are tested by every lane of the wave and (walls) hit by every ray, so the `disc >= 0`
        // skip never fires for them: the exact single-rsq sqrt (pt_core.h) instead of sqrtf()'s full expansion.  A
        // negative discriminant gives NaN roots and select_root's kMissT like the skipped form; +inf gives NaN
        // instead of +inf, and neither can beat tmin <= kMissT.
        auto test_large = [&](const float4 g, uint32_t k) {
            ++n_tests;
            const HitPre hp = intersect_pre(g.x, g.y, g.z, g.w, s.ox, s.oy, s.oz, s.dx, s.dy, s.dz);
            float q;
#if defined(__HIP_DEVICE_COMPILE__) // the sqrt variants are device-only builtins
            float am = 1.0f;
            q = sqrt_rn_rsq1(hp.disc, am);
            if (__builtin_expect(__any(am < 0x1p-96f), 0)) { // |disc| below the fast sequence's range
                asm volatile("" ::: "memory");
                q = sqrtf(hp.disc);
            }
#else
            q = sqrtf(hp.disc);
#endif
            const float t = select_root(hp.b - q, hp.b + q, ta.eps);
            if (t < tmin || (t == tmin && (int)k < idx)) { tmin = t; idx = (int)k; }
        };
        for (uint32_t i = 0; i < h.nlarge; ++i) { const uint32_t k = large[i]; test_large(geom[k], k); } // wave-uniform: scalar loads
        const float dd = s.dx * s.dx + s.dy * s.dy + s.dz * s.dz;
        const bool unit = fabsf(dd - 1.0f) <= 1e-3f; // false for NaN/inf
        if (!fin && !unit) {
            for (uint32_t k = 0; k < ns; ++k) test(k);
        } else if (!fin) {
            // slab test against the grid box; all DDA state in scalars (no indexed arrays -> no scratch)
            float tn = 0.0f, tf = 3.0e38f;
            bool inbox = true;
            // One v_rcp_f32 per axis serves the slab test and the DDA increments (1 ulp is irrelevant here: the
            // walk's exit test carries 1e-3 relative slack plus the binning margin, and a sphere near a cell corner is
            // listed in every cell its inflated box touches, whichever of two near-simultaneous crossings comes first).
            auto recip = [&](float dv) {
#if defined(__HIP_DEVICE_COMPILE__)
                return __builtin_amdgcn_rcpf(dv);
#else
                return 1.0f / dv;
#endif
            };
            const float ix = recip(s.dx), iy = recip(s.dy), iz = recip(s.dz);
            auto slab = [&](float o, float dv, float inv, float lo, float hi) {
                if (fabsf(dv) > 1e-20f) {
                    const float t1 = (lo - o) * inv, t2 = (hi - o) * inv;
                    tn = fmaxf(tn, fminf(t1, t2));
                    tf = fminf(tf, fmaxf(t1, t2));
                } else if (!(o >= lo && o <= hi)) inbox = false;
            };
            slab(s.ox, s.dx, ix, h.gmin[0], h.gmax[0]);
            slab(s.oy, s.dy, iy, h.gmin[1], h.gmax[1]);
            slab(s.oz, s.dz, iz, h.gmin[2], h.gmax[2]);
            if (inbox && tn <= tf) {
                auto axis = [&](float o, float dv, float inv, float lo, float cellw, float invw, int na, int &c, int &step,
                                float &tmax, float &tdel) {
                    int ci = (int)floorf((o + dv * tn - lo) * invw);
                    ci = ci < 0 ? 0 : (ci >= na ? na - 1 : ci);
                    c = ci;
                    if (dv > 1e-20f) { step = 1; tmax = (lo + (float)(ci + 1) * cellw - o) * inv; tdel = cellw * inv; }
                    else if (dv < -1e-20f) { step = -1; tmax = (lo + (float)ci * cellw - o) * inv; tdel = -cellw * inv; }
                    else { step = 0; tmax = 3.0e38f; tdel = 3.0e38f; }
                };
                int c0, c1, c2, st0, st1, st2;
                float tm0, tm1, tm2, td0, td1, td2;
                axis(s.ox, s.dx, ix, h.gmin[0], h.cell[0], h.inv_cell[0], n0, c0, st0, tm0, td0);
                axis(s.oy, s.dy, iy, h.gmin[1], h.cell[1], h.inv_cell[1], n1, c1, st1, tm1, td1);
                axis(s.oz, s.dz, iz, h.gmin[2], h.cell[2], h.inv_cell[2], n2, c2, st2, tm2, td2);
                const int max_steps = n0 + n1 + n2 + 3;
                uint32_t cell = (uint32_t)((c2 * n1 + c1) * n0 + c0);
                uint32_t b = cells[cell], e = cells[cell + 1];
                for (int it = 0; it < max_steps; ++it) {
                    ++n_cells;
                    n_tests += e - b;
                    // Which cell comes next depends on the crossing parameters only, not on what the candidates of
                    // this cell turn out to be: fetch its item range now, so that the dependent load is in flight
                    // while they are tested (the fetch is wasted when the walk ends here).
                    const float te = fminf(tm0, fminf(tm1, tm2)); // parameter at which the ray leaves this cell
                    const bool s0 = tm0 <= tm1 && tm0 <= tm2, s1 = !s0 && tm1 <= tm2, s2 = !s0 && !s1;
                    if (s0) { c0 += st0; tm0 += td0; }
                    if (s1) { c1 += st1; tm1 += td1; }
                    if (s2) { c2 += st2; tm2 += td2; }
                    const bool inside = (unsigned)c0 < (unsigned)n0 && (unsigned)c1 < (unsigned)n1 && (unsigned)c2 < (unsigned)n2;
                    uint32_t nb = 0, ne = 0;
                    if (inside) {
                        cell = (uint32_t)((c2 * n1 + c1) * n0 + c0);
                        nb = cells[cell];
                        ne = cells[cell + 1];
                    }
                    uint32_t i = b;
                    for (; i + 2 <= e; i += 2) { // two candidates per step: their loads are in flight together
                        const float4 ga = item_geom[i], gb = item_geom[i + 1];
                        test_item(ga, i);
                        test_item(gb, i + 1);
                    }
                    if (i < e) test_item(item_geom[i], i);
                    if (tmin < te - (1e-3f * fabsf(te) + h.margin)) break; // nothing nearer can lie ahead
                    if (!inside) break;
                    b = nb;
                    e = ne;
                }
            }
        }
        if (!fin) {
            if (pos != ~0u) idx = (int)items[pos];
            const uint32_t g = (idx < 0) ? ns - 1 : (uint32_t)idx;
            const float4 gc = geom[g];
#if defined(__HIP_DEVICE_COMPILE__)
            {   // exact fast sqrt / shared-reciprocal divide (pt_core.h); out-of-range operands redo the step with sqrtf() and '/'
                PathState n = s;
                float amin = 1.0f;
                shade_and_reflect<MODE, true>(n, tmin, gc.x, gc.y, gc.z, colx[g], coly[g], colz[g], idx == ta.light, &amin);
                if (__builtin_expect(__any(amin < 0x1p-96f), 0)) {
                    asm volatile("" ::: "memory");
                    n = s;
                    shade_and_reflect<MODE>(n, tmin, gc.x, gc.y, gc.z, colx[g], coly[g], colz[g], idx == ta.light);
                }
                s = n;
            }
#else
            shade_and_reflect<MODE>(s, tmin, gc.x, gc.y, gc.z, colx[g], coly[g], colz[g], idx == ta.light);
#endif
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


} // namespace
