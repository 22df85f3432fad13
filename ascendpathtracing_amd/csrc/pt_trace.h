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
#include "pt_leaf.h"

namespace {

using namespace apt;


constexpr int kBlock = 256;      // 4 waves
constexpr int kScene8 = 0, kSceneTiles = 1, kSceneGrid = 2; // template parameter SC: how the scene reaches the lanes
constexpr int kTile = 1024;      // spheres per LDS tile (16 KB)
#ifndef APT_GRID_WAVES
#define APT_GRID_WAVES 8 // min waves per SIMD requested for the grid-walk kernels: the walk is latency bound
                         // (dependent cell -> item loads), measured 464 / 371 / 334 / 311 / 302 ms at 3 / 4 / 5 / 6 / 8 waves
#endif
#ifndef APT_TILE_WAVES
#define APT_TILE_WAVES 6 // min waves per SIMD of the LDS-tile frame kernels (4 / 5 / 6 measured within 1 % of each other on C4 brute force)
#endif
#ifndef APT_TWO_WAVES
#define APT_TWO_WAVES 4 // min waves per SIMD of the two-paths-per-lane frame kernel (pt_trace2.h): twice the path state
#endif
#ifndef APT_FULL_WAVES
#define APT_FULL_WAVES 6 // min waves per SIMD requested for the full-trace frame kernel: caps it at 80 VGPRs (the scheduler
                         // otherwise interleaves three sphere pairs and lands on 81 -> 5 waves)
#endif
constexpr int kMaxStack = 8;
constexpr int kStackSlots = kBlock / 8; // one pairwise-sum stack per sub-pixel group (its 8 lanes hold equal values)


struct Scene8 { // wave-uniform registers (SGPRs)
    float cx[8], cy[8], cz[8], r2[8];
    bool planes; // scene8_shares_planes(), set by load_scene8()
};
// Spheres that share a centre coordinate share every term that depends on that coordinate only (c - o, its product with the
// ray direction, its square): computing such a term once is exact common-subexpression elimination, the same operation on
// the same operands.  The six walls of the reference scene are axis-aligned giant spheres (gen_data.py:94-102): spheres
// 2, 3, 4, 5 and 7 share cx, spheres 0-3 share cy, spheres 0, 1, 4, 5 and 7 share cz -- 13 distinct coordinates instead of
// 24.  A scene table with that equality pattern (compared bit for bit, per wave, from the SGPR copy; any other table takes the
// general form) is intersected through intersect_pre_planes(): 49 instead of 64 packed instructions for the 8 discriminants.
__device__ __forceinline__ bool scene8_shares_planes(const Scene8 &sc) {
    auto eq = [](float a, float b) { return f32_bits(a) == f32_bits(b); };
    return eq(sc.cx[2], sc.cx[3]) && eq(sc.cx[2], sc.cx[4]) && eq(sc.cx[2], sc.cx[5]) && eq(sc.cx[2], sc.cx[7]) &&
           eq(sc.cy[0], sc.cy[1]) && eq(sc.cy[0], sc.cy[2]) && eq(sc.cy[0], sc.cy[3]) &&
           eq(sc.cz[0], sc.cz[1]) && eq(sc.cz[0], sc.cz[4]) && eq(sc.cz[0], sc.cz[5]) && eq(sc.cz[0], sc.cz[7]);
}

// LDS table of the 8-sphere scene: entry k of `geo` = (cx, cy, cz, r2), entry k of `alb` = (albedo, 0); 16 bytes per
// entry, so that the byte offsets of entry k's index bits (16, 32, 64) are inline constants (bounce_ns8_v2).
struct Tab8 { const float4 *geo, *alb; };
constexpr int kTab8Floats4 = 17; // 8 geometry + 8 albedo entries + (1,1,1)

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
    uint32_t grid_walk;         // frame kernels: 0 = render_frame_kernel walks the grid (nested item walk); 2 = it returns at once when
                                // the sample-queue kernel's grid form renders this frame (grid_queue_usable); sample-queue kernel:
                                // 3 = APT_FLAG_GRID_SLOTS, it is the frame's only launch and reports a grid it cannot use
    unsigned long long *traced; // optional device counter of traced segments
    uint32_t *status;           // the context's device status word (include/render_mi355x.h APT_DEV_*), or null
};
// Workgroup id -> the block of work it takes, XCD-aware.  The dispatcher hands consecutive workgroup ids to the 8 XCDs of an MI355X in turn,
// and each XCD has its own L2: with the identity mapping the workgroups that write one 128-byte line of the frame (32 float pixels of a plane,
// 42 u8 pixels) sit on different XCDs, every L2 holds -- and writes back -- a PART of the line, and the fabric saw 1.20x the frame's bytes
// (rounds 2-4: 37.3 MB written per C2 launch for 31.1, steady state).  Here the blocks are taken in chunks of 8 * K: within a chunk XCD x takes K
// CONSECUTIVE blocks (its j-th workgroup of the chunk takes block x * K + j), so neighbouring pixels are written through one L2 at about the same
// time and leave it as whole lines -- 31.10 MB written, every byte once --, while every XCD still works on an interleaved 1/8 sample of the whole
// frame.  (The first form gave XCD x the x-th contiguous EIGHTH of the frame: same traffic, but where the cost per pixel varies over the image --
// retirement, roulette, the grid scenes -- the XCDs then finish far apart: C2 with retirement 15.7 -> 18.3 ms, depth 32 33.3 -> 42.4, C4 +29 %;
// profiles/r05_variants_xcd_mapping.jsonl.)  The last, partial chunk keeps the identity.  Pure relabelling of independent blocks: no effect on
// any result.
template <uint32_t K>
__device__ __forceinline__ uint32_t xcd_chunked_block(uint32_t wg, uint32_t nwg) {
    constexpr uint32_t kXcds = 8, G = kXcds * K;
    static_assert((K & (K - 1)) == 0, "shifts and masks");
    const uint32_t full = nwg & ~(G - 1u);
    if (wg >= full) return wg;
    const uint32_t r = wg & (G - 1u);
    return (wg & ~(G - 1u)) + (r & (kXcds - 1u)) * K + r / kXcds;
}

// A kernel that has to give up says so (the reference asserts inside its kernel: src/render.cpp:68-73): one lane ORs the bit into the
// context's status word; apt_context_check() reports it as APT_ERR_DEVICE.
__device__ __forceinline__ void report_status(const TraceArgs &ta, uint32_t bit) {
    if (ta.status && (threadIdx.x & 63u) == 0u) atomicOr(ta.status, bit);
}

// Discriminants of TWO spheres per instruction: the tile is stored as sphere pairs,
//   tile[2p]   = (cx[2p], cx[2p+1], cy[2p], cy[2p+1])      tile[2p+1] = (cz[2p], cz[2p+1], r2[2p], r2[2p+1])
// so every operation of intersect_pre becomes one v_pk_{add,mul}_f32 over a register pair, the
// ray component being broadcast to both halves by op_sel (no register shuffles).  Packed fp32
// ops round exactly like the scalar ones, element by element; contraction is off.

// The "0 +" in front of the shading step's dot product (rt_helper.h:690 Duplicate(0); sdot's `dot = 0.0` in O-mode) only matters when
// all three products are -0: the sum is then +0 with it and -0 without.  The one consumer is k2 = dot * 2 (:697), and fma(dot, 2, +0)
// is that product exactly (a doubling never rounds; overflow and NaN go the same way) with -0 turned into +0 by the addend: the start
// value's whole effect for the price of the multiply itself.  (The norm's products are squares, never -0: nothing to preserve there.)
__device__ __forceinline__ float twice_canonical(float dot) { return __builtin_fmaf(dot, 2.0f, 0.0f); }
__device__ __forceinline__ f2 twice_canonical(f2 dot) { return __builtin_elementwise_fma(dot, f2{2.0f, 2.0f}, f2{0.0f, 0.0f}); }

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


// The 8 discriminants of a scene that passed scene8_shares_planes(), sphere pairs (0,1) (2,3) (4,5) (6,7) as intersect_pre2
// returns them.  Distinct coordinates in register pairs:  X1 = (cx0, cx1)  X2 = (cx2, cx6)   Y1 = (cy4, cy5)  Y2 = (cy6, cy7)
// Y3 = (cy0, cy0)   Z1 = (cz2, cz3)  Z2 = (cz0, cz6); every sum of a sphere pair finds both of its operands inside ONE pair per
// source (op_sel picks the halves), so nothing is shuffled.  Operation for operation what intersect_pre2 computes per sphere.
__device__ __forceinline__ void intersect_pre_planes_pairs(const f2 X1, const f2 X2, const f2 Y1, const f2 Y2, const f2 Y3, const f2 Z1, const f2 Z2,
                                                           const f2 R0, const f2 R1, const f2 R2, const f2 R3, float ox, float oy, float oz,
                                                           float dx, float dy, float dz, HitPre2 (&h)[4]) {
    const f2 x1 = X1 - ox, x2 = X2 - ox, y1 = Y1 - oy, y2 = Y2 - oy, y3 = Y3 - oy, z1 = Z1 - oz, z2 = Z2 - oz;
    auto lo = [](f2 v) { return f2{v.x, v.x}; };
    auto swap = [](f2 v) { return f2{v.y, v.x}; };
    {   // b = (ocx*dx + ocy*dy) + ocz*dz
        const f2 px1 = x1 * dx, px2 = x2 * dx, py1 = y1 * dy, py2 = y2 * dy, py3 = y3 * dy, pz1 = z1 * dz, pz2 = z2 * dz;
        h[0].b = (px1 + py3) + lo(pz2);
        h[1].b = (lo(px2) + py3) + pz1;
        h[2].b = (lo(px2) + py1) + lo(pz2);
        h[3].b = (swap(px2) + py2) + swap(pz2);
    }
    {   // c = ((ocx^2 + ocy^2) + ocz^2) - r2;  disc = b*b - c
        const f2 qx1 = x1 * x1, qx2 = x2 * x2, qy1 = y1 * y1, qy2 = y2 * y2, qy3 = y3 * y3, qz1 = z1 * z1, qz2 = z2 * z2;
        f2 c0 = (qx1 + qy3) + lo(qz2), c1 = (lo(qx2) + qy3) + qz1, c2 = (lo(qx2) + qy1) + lo(qz2), c3 = (swap(qx2) + qy2) + swap(qz2);
        c0 = c0 - R0; c1 = c1 - R1; c2 = c2 - R2; c3 = c3 - R3;
        h[0].disc = h[0].b * h[0].b - c0; h[1].disc = h[1].b * h[1].b - c1;
        h[2].disc = h[2].b * h[2].b - c2; h[3].disc = h[3].b * h[3].b - c3;
    }
}
__device__ __forceinline__ void intersect_pre_planes(const Scene8 &sc, float ox, float oy, float oz, float dx, float dy, float dz,
                                                     HitPre2 (&h)[4]) {
    intersect_pre_planes_pairs(f2{sc.cx[0], sc.cx[1]}, f2{sc.cx[2], sc.cx[6]}, f2{sc.cy[4], sc.cy[5]}, f2{sc.cy[6], sc.cy[7]}, f2{sc.cy[0], sc.cy[0]},
                               f2{sc.cz[2], sc.cz[3]}, f2{sc.cz[0], sc.cz[6]}, f2{sc.r2[0], sc.r2[1]}, f2{sc.r2[2], sc.r2[3]}, f2{sc.r2[4], sc.r2[5]},
                               f2{sc.r2[6], sc.r2[7]}, ox, oy, oz, dx, dy, dz, h);
}

// ---- trace: reference scene (Ns == 8) ----------------------------------------------------
// One bounce in the reference's own float form: 8 intersections (sphere operands in SGPRs) with sqrtf(), select_root's two
// selects and the strict-'<' arg-min, gather, shade with sqrtf() and '/'.  The COLD form: what a wave falls back to when a lane
// leaves the validity range of the fast sequences below, or when eps does not permit the integer root keys.
template <int MODE>
__device__ __forceinline__ void bounce_ns8_exact(const Scene8 &sc, const Tab8 tab, const PathState &s, PathState &n,
                                                 const TraceArgs &ta) {
    const int miss = (MODE == kModeOracle) ? -1 : 0; // all-miss: gen_data.py:311 / rt_helper.h:183-201
    float tmin = kMissT;
    int idx = miss;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float t0, t1;
        intersect_roots(sc.cx[k], sc.cy[k], sc.cz[k], sc.r2[k], s.oxy.x, s.oxy.y, s.oz, s.dxy.x, s.dxy.y, s.dz, t0, t1);
        const float t = select_root(t0, t1, ta.eps);
        if (t < tmin) { tmin = t; idx = k; } // strict '<', ascending k: lowest index wins ties
    }
    const int g = (idx < 0) ? 7 : idx; // Python index -1 wraps to the last sphere
    const float4 c = tab.geo[g], col = tab.alb[g];
    n = s;
    shade_and_reflect<MODE>(n, tmin, c.x, c.y, c.z, col.x, col.y, col.z, idx == ta.light);
}

// ---- the hot form of that bounce ---------------------------------------------------------------------------
// On gfx950 a wave64 VALU instruction issues in ~2.5 cycles when it is a plain fp32/int32 operation on VGPRs
// and in ~4.3 cycles when it is a compare, select, min/max, or has an SGPR operand (profiles/microbench/
// valu_rates_mi355x.txt); the bounce block is VALU-issue bound, so its time is the sum of those costs.  Same
// arithmetic as bounce_ns8_exact in the fp32 data path, with the exact fast sqrt sequence (pt_core.h sqrt_rn_rsq1) and the
// integer root keys (pt_core.h, "the same selection ... in the integer domain"); how it is written:
//   * the root-key bias and start value live in VGPRs (KeyConsts): the 16 integer subtractions per bounce no
//     longer carry an SGPR operand (half rate -> full rate);
//   * the arg-min index is not tracked per lane with compare + select per sphere: the 8 "sphere k improved the
//     minimum" compares write wave masks, three scalar accumulators collect the bits of the index on the SALU
//     (which issues beside the VALU), and three selects + one or3 turn them into the LDS address of the hit
//     sphere (16 bytes per entry: 16, 32 and 64 are inline constants);
//   * the validity tracking of the fast sqrt / divide sequences is v_min3_f32 with |.| source modifiers: one
//     instruction per sphere pair, two in the shading step;
//   * the throughput update runs under the alive mask (exec) instead of through three selects;
//   * "does any lane need the exact re-run" is the OR of two wave masks, not a materialised boolean.
struct KeyConsts { uint32_t bias, init; uint64_t nbias2; };
__device__ __forceinline__ KeyConsts make_key_consts(float eps) {
    KeyConsts kc;
    kc.bias = f32_bits(eps) + 1u;
    kc.init = f32_bits(kMissT) - kc.bias;
    // keys of a (-t0, t1) register pair by ONE 64-bit add (root_pair / intersect_ns8_v2)
    kc.nbias2 = ((uint64_t)(0u - kc.bias) << 32) | (0x80000000u - kc.bias);
    asm volatile("" : "+v"(kc.bias), "+v"(kc.init)); // opaque: the compiler must keep them in VGPRs
    asm volatile("" : "+s"(kc.nbias2));
    return kc;
}
// (-t0, t1) = (q - b, b + q) of ONE sphere as the two halves of a register pair, from the packed (sphere k, sphere k+1)
// operands: v_pk_add_f32 with both result halves reading half `HALF` of the sources and the low half negating b
// (exact: round-to-nearest is symmetric, q - b is -(b - q) bit for bit).
template <int HALF>
__device__ __forceinline__ uint64_t root_pair(f2 b, f2 q) {
    uint64_t r;
    if (HALF == 0) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_lo:[1,0] neg_hi:[0,0]" : "=v"(r) : "v"(b), "v"(q));
    else asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,0]" : "=v"(r) : "v"(b), "v"(q));
    return r;
}
__device__ __forceinline__ float min3_abs(float a, float b, float c) { // min(a, |b|, |c|); NaNs drop out (IEEE minNum)
    float r;
    asm("v_min3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// The validity chain of the v2 bounce: min(a, |b|, |c|) that PROPAGATES NaN (v_minimum3_f32, IEEE 754-2019 minimum), so that
// a NaN operand stays flagged, against ONE threshold kFastMin = 2^-29 for everything the fast sequences need:
//   |discriminant| (sqrt_rn_rsq1 needs >= 2^-96), |nx|, |ny|, |nz| (div3 numerators: not -0, >= 2^-96; len2 >= 2^-58 then needs
//   no test of its own), and rsq(len2) in place of len2 <= 2^60 (pt_core.h div3_operands_ok: rsq(len2) >= 2^-29 is len2 <= 2^58
//   up to the rsq's last ulp, well inside 2^60; +inf / NaN give 0 / NaN).  A path that fails redoes the bounce exactly, so a
//   stricter test only costs time: |component| < 2^-29 has probability ~1e-10 per bounce in these scenes.
constexpr float kFastMin = 0x1p-29f;
__device__ __forceinline__ float minimum3_abs(float a, float b, float c) {
    float r;
    asm("v_minimum3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// The same when an operand was just written by a transcendental instruction (v_rsq_f32): gfx950 needs one wait state between
// a TRANS result and a non-TRANS VALU instruction that reads it.  The compiler inserts it for its own instructions but does not
// look into inline asm (measured: without the s_nop the minimum read the register's previous contents and ~2 % of the waves of
// the sample-queue kernels took the exact path for nothing -- still bit-exact, the exact path always is).
__device__ __forceinline__ float minimum3_abs_after_trans(float a, float b, float c) {
    float r;
    asm("s_nop 0\n\tv_minimum3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t min3_u32(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t select_const(uint64_t mask, uint32_t value_if_set) { // value must be an inline constant
    uint32_t r;
    asm("" : "+s"(mask)); // a mask the compiler knows to be constant must still arrive in an SGPR pair
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "n"(value_if_set), "s"(mask));
    return r;
}

// -> wave mask of the lanes that left the validity range of the fast sequences (exact re-run wanted)
// `alive`: wave mask of the lanes whose path has not reached the light (in: before, out: after this bounce);
// s.alive / n.alive are not read or written here (callers that need the per-lane form make it from the mask).
struct Albedo { f2 xy; float z; };
// ret *= albedo for the lanes of `alive`, in place: the three products run under exec = alive (an s_and_saveexec /
// restore pair on the scalar unit) instead of through three selects; other lanes keep their value (x1 is exact).
__device__ __forceinline__ void apply_albedo(f2 &rxy, float &rz, const Albedo &a, uint64_t alive) {
    uint64_t saved;
    asm("" : "+s"(alive)); // see select_const
    asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                 "v_pk_mul_f32 %[rxy], %[cxy], %[rxy]\n\t"
                 "v_mul_f32 %[rz], %[cz], %[rz]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [sv] "=&s"(saved), [rxy] "+v"(rxy), [rz] "+v"(rz)
                 : [m] "s"(alive), [cxy] "v"(a.xy), [cz] "v"(a.z)
                 : "scc");   // s_and_saveexec writes SCC: the compiler must not keep a scalar compare alive across this
}

// The intersection half of a bounce: all 8 spheres against one ray, integer-key arg-min, -> nearest accepted root,
// byte offset of the hit sphere's LDS table entry (index * 16) and the wave mask of the lanes that hit the light.
struct Hit8 { float tmin; uint32_t addr; uint64_t light; };
template <int MODE, bool PLANES = false>
__device__ __forceinline__ Hit8 intersect_ns8_v2(const Scene8 &sc, float ox, float oy, float oz, float dx, float dy, float dz,
                                                 const TraceArgs &ta, const KeyConsts &kc, float &amin) {
    uint32_t best = kc.init;
    uint64_t b0 = 0, b1 = 0, b2 = 0, any = 0;
    auto update_keys = [&](uint32_t m0, uint32_t m1, int k) {
        const uint32_t nb = min3_u32(best, m0, m1);
        const uint64_t better = __builtin_amdgcn_ballot_w64(nb != best); // strict '<': lowest index wins ties
        best = nb;
        b0 = (k & 1) ? (b0 | better) : (b0 & ~better);
        b1 = (k & 2) ? (b1 | better) : (b1 & ~better);
        b2 = (k & 4) ? (b2 | better) : (b2 & ~better);
        if (MODE == kModeOracle) any |= better;
    };
    auto update64 = [&](uint64_t keys, int k) { update_keys((uint32_t)keys, (uint32_t)(keys >> 32), k); };
    HitPre2 hp[4];
    if (PLANES) intersect_pre_planes(sc, ox, oy, oz, dx, dy, dz, hp);
#pragma unroll
    for (int k = 0; k < 8; k += 2) { // rt_helper.h:457-467, two spheres per packed instruction
        const HitPre2 h = PLANES ? hp[k / 2]
                                 : intersect_pre2(f2{sc.cx[k], sc.cx[k + 1]}, f2{sc.cy[k], sc.cy[k + 1]}, f2{sc.cz[k], sc.cz[k + 1]},
                                                  f2{sc.r2[k], sc.r2[k + 1]}, ox, oy, oz, dx, dy, dz);
        amin = minimum3_abs(amin, h.disc.x, h.disc.y);
        // sqrt_rn_rsq1 on both lanes of the pair (pt_core.h): y = x*r, hh = r/2, q = fma(fma(-y,y,x), hh, y)
        const f2 r0 = {__builtin_amdgcn_rsqf(h.disc.x), __builtin_amdgcn_rsqf(h.disc.y)};
        const f2 y = h.disc * r0, hh = r0 * 0.5f;
        const f2 res = __builtin_elementwise_fma(-y, y, h.disc);
        const f2 q = __builtin_elementwise_fma(res, hh, y);
        // Both keys of a sphere by ONE 64-bit add of (2^31 - bias, -bias) to its (-t0, t1) register pair (u(x) = bit pattern).
        // Low half: u(-t0) + 2^31 - bias (mod 2^32) is key(t0) = u(t0) - bias when t0 > eps (u(-t0) = 2^31 + u(t0)); for every other
        // t0 -- negative: u(|t0|) + 2^31 - bias; in [+0, eps]: 2^32 - (bias - u(t0)); NaN, inf -- it is >= key(kMissT), like the
        // 32-bit form's invalid keys, so it can never win.  The low half carries into the high half exactly when t0 > eps (or is a
        // sign-bit NaN / +inf, where t1 is NaN / inf too): the high half is then key(t1) + 1, but t1 >= t0 > eps, so key(t0) <=
        // key(t1) < key(t1) + 1 with no wrap (key(t1) = 2^32 - 1 would mean t1 = eps), and min3 returns key(t0) either way.
        // `best` therefore only ever holds a true key or the initial one, and tmin = value(best) as before.
        update64(root_pair<0>(h.b, q) + kc.nbias2, k);
        update64(root_pair<1>(h.b, q) + kc.nbias2, k + 1);
    }
    const float tmin = bits_f32(best + kc.bias);
    uint64_t light_mask; // lanes whose arg-min is the light
    if (MODE == kModeOracle) { // all-miss: Python index -1 = the last sphere's geometry and colour, never "the light" (gen_data.py:311,343,390)
        b0 |= ~any; b1 |= ~any; b2 |= ~any;
    }
    uint32_t addr;
    {
        const uint32_t a0 = select_const(b0, 16), a1 = select_const(b1, 32), a2 = select_const(b2, 64);
        asm("v_or3_b32 %0, %1, %2, %3" : "=v"(addr) : "v"(a0), "v"(a1), "v"(a2));
    }
    // (Deriving this mask from b0 / b1 / b2 on the SALU saves the compare but costs 6-8 more spilled SGPRs in every kernel:
    // 21.89 against 21.84 ms at C2, +3 % on the sample-queue kernels -- measured in round 2, not kept.)
    light_mask = __builtin_amdgcn_ballot_w64(addr == (uint32_t)ta.light * 16u); // light < 0 or > 7 never matches
    if (MODE == kModeOracle) light_mask &= any;
    return Hit8{tmin, addr, light_mask};
}

template <int MODE, bool PLANES = false>
__device__ __forceinline__ uint64_t bounce_ns8_v2(const Scene8 &sc, const Tab8 tab, const PathState &s, PathState &n,
                                                  const TraceArgs &ta, const KeyConsts &kc, uint64_t &alive, Albedo &albedo) {
    float amin = 1.0f;
    const Hit8 hit = intersect_ns8_v2<MODE, PLANES>(sc, s.oxy.x, s.oxy.y, s.oz, s.dxy.x, s.dxy.y, s.dz, ta, kc, amin);
    const float tmin = hit.tmin;
    const uint32_t addr = hit.addr;
    const uint64_t light_mask = hit.light;
    const float4 c = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(tab.geo) + addr);
    const float4 col = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(tab.alb) + addr);
    // GenerateNewRays (rt_helper.h:504-709), as shade_and_reflect<MODE, true>, with the x and y components of
    // every 3-vector operation in ONE packed instruction (v_pk_{mul,add,fma}_f32 round each half exactly like the
    // scalar instruction; nothing is contracted): in this kernel every VALU instruction costs about one 4-cycle
    // issue slot whatever its class (measured in place: profiles/history/r02_insitu_costs.md), so the instruction COUNT
    // is what the bounce costs and a packed pair is two operations for one slot.
    const f2 oxy = s.oxy, dxy = s.dxy;
    const f2 hxy = oxy + dxy * tmin;                           // :513-518  h = o + d*t (mul, then add)
    const float hz = s.oz + s.dz * tmin;
    const f2 nxy = hxy - f2{c.x, c.y};                         // :635-637
    const float nz = hz - c.z;
    const f2 sq = nxy * nxy;
    float len2;
    if (MODE == kModeOracle) {                                 // np.linalg.norm, gen_data.py:347: float64 accumulation
        const float p2 = nz * nz;
        double acc = (double)sq.x;                             // (sdot starts from 0.0: 0 + a square is that square)
        acc = acc + (double)sq.y;
        acc = acc + (double)p2;
        len2 = (float)acc;
    } else {
        float acc = sq.x + sq.y;                               // :641-649 (0 + x^2 is x^2: a square is never -0)
        acc = acc + nz * nz;
        len2 = acc;
    }
    float L;
    {
        const float r0 = __builtin_amdgcn_rsqf(len2);
        amin = minimum3_abs_after_trans(amin, r0, nxy.x); // validity of the fast sqrt / divide sequences: see kFastMin
        amin = minimum3_abs(amin, nxy.y, nz);
        const float y = len2 * r0, h = 0.5f * r0;
        const float r = __builtin_fmaf(-y, y, len2);
        L = __builtin_fmaf(r, h, y);
    }
    f2 uxy;
    float uz;
    div3_packed(nxy, nz, L, uxy, uz);           // pt_core.h; validity = the amin / huge tests of this function
    const f2 pr = dxy * uxy;
    const float pz = s.dz * uz;
    float dot;
    if (MODE == kModeOracle) {                                 // np.dot, gen_data.py:349
        double acc = (double)pr.x;                             // (sdot's 0.0 start: twice_canonical() below)
        acc = acc + (double)pr.y;
        acc = acc + (double)pz;
        dot = (float)acc;
    } else {
        dot = pr.x;                                            // :690 Duplicate(0): twice_canonical() below; :694-696
        dot = dot + pr.y;
        dot = dot + pz;
    }
    const float k2 = twice_canonical(dot);                    // :697
    n.dxy = dxy - uxy * k2;                                    // :699-704
    n.dz = s.dz - uz * k2;
    n.oxy = hxy; n.oz = hz;                                    // :706-708
    // AccumulateIntervalColor (rt_helper.h:711-830): alive &= idx != light; ret *= alive ? albedo : 1.  The mask
    // is updated here; the multiplication itself is apply_albedo(), which the caller runs once it knows that this
    // bounce stands (no exact re-run): it can then overwrite the throughput registers in place.
    alive &= ~light_mask;
    n.alive = s.alive;
    albedo = Albedo{f2{col.x, col.y}, col.z};
    return __builtin_amdgcn_ballot_w64(!(amin >= kFastMin)); // something too small, len2 > 2^60, or a NaN
}

// bounce_ns8_v2 in TWO phases, for a loop that keeps ONE set of state registers (pt_queue.h): everything the validity test of the
// fast sequences needs is known after the hit point, the normal and rsq(|normal|^2) -- BEFORE anything of the new ray is written.
// A caller that branches to its exact form between the phases still holds the untouched ray there, and on the fast path the new ray
// can be written over the old one (its last reads are the instructions that produce the new values): no copies at the back edge.
// Same operations in the same order as bounce_ns8_v2.
struct Bounce8Mid {            // what phase 2 needs from phase 1
    f2 hxy, nxy;
    float hz, nz, len2, r0;
    float4 col;
    uint64_t light;
};
template <int MODE, bool PLANES>
__device__ __forceinline__ uint64_t bounce_ns8_v2_hit(const Scene8 &sc, const Tab8 tab, const PathState &s, const TraceArgs &ta, const KeyConsts &kc,
                                                      Bounce8Mid &m) {
    float amin = 1.0f;
    const Hit8 hit = intersect_ns8_v2<MODE, PLANES>(sc, s.oxy.x, s.oxy.y, s.oz, s.dxy.x, s.dxy.y, s.dz, ta, kc, amin);
    const float tmin = hit.tmin;
    const float4 c = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(tab.geo) + hit.addr);
    m.col = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(tab.alb) + hit.addr);
    m.light = hit.light;
    m.hxy = s.oxy + s.dxy * tmin;                             // :513-518  h = o + d*t (mul, then add)
    m.hz = s.oz + s.dz * tmin;
    m.nxy = m.hxy - f2{c.x, c.y};                              // :635-637
    m.nz = m.hz - c.z;
    const f2 sq = m.nxy * m.nxy;
    if (MODE == kModeOracle) {                                 // np.linalg.norm, gen_data.py:347: float64 accumulation
        const float p2 = m.nz * m.nz;
        double acc = (double)sq.x;                             // (sdot starts from 0.0: 0 + a square is that square)
        acc = acc + (double)sq.y;
        acc = acc + (double)p2;
        m.len2 = (float)acc;
    } else {
        float acc = sq.x + sq.y;                               // :641-649 (0 + x^2 is x^2: a square is never -0)
        acc = acc + m.nz * m.nz;
        m.len2 = acc;
    }
    m.r0 = __builtin_amdgcn_rsqf(m.len2);
    amin = minimum3_abs_after_trans(amin, m.r0, m.nxy.x);      // validity of the fast sqrt / divide sequences: see kFastMin
    amin = minimum3_abs(amin, m.nxy.y, m.nz);
    return __builtin_amdgcn_ballot_w64(!(amin >= kFastMin));   // something too small, len2 > 2^60, or a NaN
}
template <int MODE>
__device__ __forceinline__ void bounce_ns8_v2_reflect(PathState &s, const Bounce8Mid &m, uint64_t &alive, Albedo &albedo) {
    float L;
    {
        const float y = m.len2 * m.r0, h = 0.5f * m.r0;
        const float r = __builtin_fmaf(-y, y, m.len2);
        L = __builtin_fmaf(r, h, y);
    }
    f2 uxy;
    float uz;
    div3_packed(m.nxy, m.nz, L, uxy, uz);
    const f2 pr = s.dxy * uxy;
    const float pz = s.dz * uz;
    float dot;
    if (MODE == kModeOracle) {                                 // np.dot, gen_data.py:349
        double acc = (double)pr.x;                             // (sdot's 0.0 start: twice_canonical() below)
        acc = acc + (double)pr.y;
        acc = acc + (double)pz;
        dot = (float)acc;
    } else {
        dot = pr.x;                                            // :690 Duplicate(0): twice_canonical() below; :694-696
        dot = dot + pr.y;
        dot = dot + pz;
    }
    const float k2 = twice_canonical(dot);                    // :697
    s.dxy = s.dxy - uxy * k2;                                  // :699-704
    s.dz = s.dz - uz * k2;
    s.oxy = m.hxy; s.oz = m.hz;                                // :706-708
    alive &= ~m.light;
    albedo = Albedo{f2{m.col.x, m.col.y}, m.col.z};
}

// The shading half of bounce_ns8_v2 for a hit given by value (nearest root tmin, centre of the hit sphere): GenerateNewRays
// (rt_helper.h:504-709) with the x and y components of every 3-vector operation in one packed instruction, the single-rsq square root and
// the shared-reciprocal divide.  Writes the new ray to n.oxy / n.oz / n.dxy / n.dz and returns the validity minimum of the fast sequences
// (see kFastMin): the caller redoes a bounce whose minimum is not >= kFastMin with shade_and_reflect<MODE>() (sqrtf() and '/').
// Operation for operation what shade_and_reflect<MODE, true>() computes for the ray.  (Grid form of the sample-queue kernel, pt_queue.h.)
template <int MODE>
__device__ __forceinline__ float reflect_packed(const PathState &s, float tmin, float cx, float cy, float cz, PathState &n) {
    const f2 hxy = s.oxy + s.dxy * tmin;                       // :513-518  h = o + d*t (mul, then add)
    const float hz = s.oz + s.dz * tmin;
    const f2 nxy = hxy - f2{cx, cy};                           // :635-637
    const float nz = hz - cz;
    const f2 sq = nxy * nxy;
    float len2;
    if (MODE == kModeOracle) {                                 // np.linalg.norm, gen_data.py:347: float64 accumulation
        const float p2 = nz * nz;
        double acc = (double)sq.x;                             // (sdot starts from 0.0: 0 + a square is that square)
        acc = acc + (double)sq.y;
        acc = acc + (double)p2;
        len2 = (float)acc;
    } else {
        float acc = sq.x + sq.y;                               // :641-649 (0 + x^2 is x^2: a square is never -0)
        acc = acc + nz * nz;
        len2 = acc;
    }
    const float r0 = __builtin_amdgcn_rsqf(len2);
    float amin = minimum3_abs_after_trans(1.0f, r0, nxy.x);    // validity of the fast sqrt / divide sequences: see kFastMin
    amin = minimum3_abs(amin, nxy.y, nz);
    float L;
    {
        const float y = len2 * r0, h = 0.5f * r0;
        const float r = __builtin_fmaf(-y, y, len2);
        L = __builtin_fmaf(r, h, y);
    }
    f2 uxy;
    float uz;
    div3_packed(nxy, nz, L, uxy, uz);
    const f2 pr = s.dxy * uxy;
    const float pz = s.dz * uz;
    float dot;
    if (MODE == kModeOracle) {                                 // np.dot, gen_data.py:349
        double acc = (double)pr.x;                             // (sdot's 0.0 start: twice_canonical() below)
        acc = acc + (double)pr.y;
        acc = acc + (double)pz;
        dot = (float)acc;
    } else {
        dot = pr.x;                                            // :690 Duplicate(0): twice_canonical() below; :694-696
        dot = dot + pr.y;
        dot = dot + pz;
    }
    const float k2 = twice_canonical(dot);                    // :697
    n.dxy = s.dxy - uxy * k2;                                  // :699-704
    n.dz = s.dz - uz * k2;
    n.oxy = hxy; n.oz = hz;                                    // :706-708
    return amin;
}

// The same with the plane-sharing form of the intersections chosen at run time (scene8_shares_planes(), wave-uniform)
template <int MODE>
__device__ __forceinline__ uint64_t bounce_ns8_v2p(const Scene8 &sc, const Tab8 tab, const PathState &s, PathState &n, const TraceArgs &ta,
                                                   const KeyConsts &kc, uint64_t &alive, Albedo &albedo, bool planes) {
    return planes ? bounce_ns8_v2<MODE, true>(sc, tab, s, n, ta, kc, alive, albedo)
                  : bounce_ns8_v2<MODE, false>(sc, tab, s, n, ta, kc, alive, albedo);
}

// One bounce of a wave through the fast form, falling back to the exact form (sqrtf, '/', float selects) for the
// whole wave when a lane whose path can still reach an output left the fast sequences' validity range, or when
// eps does not permit the integer root keys.  `alive` as in bounce_ns8_v2; s.alive is refreshed from it only on
// the cold path, n.alive is valid on return only if `want_lane_alive`.
template <int MODE>
__device__ __forceinline__ void bounce_ns8_checked(const Scene8 &sc, const Tab8 tab, const PathState &s, PathState &n,
                                                   const TraceArgs &ta, const KeyConsts &kc, bool fast_ok,
                                                   uint64_t &alive, bool want_lane_alive, uint64_t active = ~0ull) {
    const uint64_t alive_in = alive;
    uint64_t redo = ~0ull;
    Albedo albedo;
    bool fast_stands = false;
    if (__builtin_expect(fast_ok, 1)) { redo = bounce_ns8_v2p<MODE>(sc, tab, s, n, ta, kc, alive, albedo, sc.planes); fast_stands = true; }
    if (__builtin_expect((redo & active) != 0, 0)) {
        // A lane left the validity range of the fast sequences (|sqrt argument| < 2^-96, divide operands outside
        // [2^-40, 2^40]).  A lane whose path is already finished (alive bit cleared or throughput zero) cannot
        // influence any output any more, so its request is ignored: deep all-miss paths (|n| ~ 1e20) are of that
        // kind.  Otherwise redo the bounce with sqrtf() and '/'.  The empty volatile asm keeps this cold path out
        // of the hot block.
        PathState cold = s;
        cold.alive = select_const(alive_in, 1);
        const bool lane = select_const(redo & active, 1) != 0 && (!fast_ok || !path_finished(cold));
        if (__any(lane)) {
            asm volatile("" ::: "memory");
            bounce_ns8_exact<MODE>(sc, tab, cold, n, ta);
            alive = __builtin_amdgcn_ballot_w64(n.alive != 0);
            fast_stands = false;
            if (ta.traced && (threadIdx.x & 63) == 0) atomicAdd(ta.traced + 3, 1ull); // statistics: exact re-runs
        }
    }
    if (__builtin_expect(fast_stands, 1)) {
        n.rxy = s.rxy; n.rz = s.rz;
        apply_albedo(n.rxy, n.rz, albedo, alive);
    }
    if (want_lane_alive) n.alive = select_const(alive, 1);
}

template <int MODE, bool RETIRE>
__device__ __forceinline__ uint32_t trace_ns8(const Scene8 &sc, const Tab8 tab, PathState &s, bool valid,
                                              const TraceArgs &ta, uint64_t path) {
    uint32_t traced = 0;
    const uint64_t rr_key = ta.rr_start ? rr_path_key(ta.seed, path) : 0;
    const KeyConsts kc = make_key_consts(ta.eps);
    const bool fast_ok = eps_allows_rootkey(ta.eps);
    const bool lane_alive = RETIRE || ta.rr_start != 0;            // who needs s.alive per lane (wave-uniform)
    uint64_t alive = __builtin_amdgcn_ballot_w64(s.alive != 0);
    if (RETIRE) {
        for (uint32_t d = 0; d < ta.depth; ++d) { // render.cpp:140-188
            const bool fin = !valid || path_finished(s);
            if (__all(fin)) break;
            PathState n;
            uint64_t alive_n = alive;
            bounce_ns8_checked<MODE>(sc, tab, s, n, ta, kc, fast_ok, alive_n, true);
            if (ta.rr_start && d + 1 >= ta.rr_start) russian_roulette(n, rr_key, d); // wave-uniform branch
            if (!fin) { s = n; ++traced; }
            alive = __builtin_amdgcn_ballot_w64(s.alive != 0);
        }
    } else {
        // Full trace (lanes past the end of the range compute garbage that is never stored).  The hot loop contains
        // no merge of "fast result" and "exact re-run result" (the register allocator paid for that merge with a
        // dozen v_mov per bounce on the fast edge): when a bounce must be redone exactly, the wave leaves the loop
        // and finishes the path's remaining bounces in the exact form (trace_rest_exact) -- about 1e-5 of the
        // wave-bounces on the demo scene.  Two bounces per loop turn, written out: the second reads the registers
        // the first wrote and writes the first's inputs, so the new state is not copied back either (the compiler
        // cannot unroll this loop itself, it contains convergent operations).
        f2 rxy = s.rxy;   // the throughput lives outside the ping-pong pair and is updated in place
        float rz = s.rz;
        auto rest_exact = [&](PathState &from, uint32_t d) { // -> result in s
            from.alive = select_const(alive, 1);
            from.rxy = rxy; from.rz = rz;
            for (; d < ta.depth; ++d) {
                PathState n;
                bounce_ns8_exact<MODE>(sc, tab, from, n, ta);
                if (ta.rr_start && d + 1 >= ta.rr_start) russian_roulette(n, rr_key, d);
                from = n;
            }
            if (ta.traced && (threadIdx.x & 63) == 0) atomicAdd(ta.traced + 3, 1ull); // statistics: waves that left the fast loop
            s = from;
        };
        // one fast bounce in -> out (ray only); true when the wave has to go exact from `in` (out is then meaningless)
        auto step = [&](PathState &in, PathState &out, uint32_t d) -> bool {
            Albedo albedo;
            uint64_t alive_out = alive;
            const uint64_t redo = bounce_ns8_v2p<MODE>(sc, tab, in, out, ta, kc, alive_out, albedo, sc.planes);
            if (__builtin_expect(redo != 0, 0)) {
                // A lane left the validity range of the fast sequences (|sqrt argument| < 2^-96, divide operands
                // outside [2^-40, 2^40]).  A lane whose path is already finished (alive bit cleared or throughput
                // zero) cannot influence any output any more, so its request is ignored: deep all-miss paths
                // (|n| ~ 1e20) are of that kind.
                const bool fin = select_const(alive, 1) == 0 || (rxy.x == 0.0f && rxy.y == 0.0f && rz == 0.0f);
                if (__any(select_const(redo, 1) != 0 && !fin)) return true;
            }
            apply_albedo(rxy, rz, albedo, alive_out);   // in place once the bounce stands
            alive = alive_out;
            if (ta.rr_start && d + 1 >= ta.rr_start) { // wave-uniform
                PathState t;
                t.rxy = rxy; t.rz = rz; t.alive = select_const(alive, 1);
                russian_roulette(t, rr_key, d);
                rxy = t.rxy; rz = t.rz;
            }
            return false;
        };
        traced = ta.depth;
        PathState n;
        uint32_t d = 0;
        if (__builtin_expect(!fast_ok, 0)) { rest_exact(s, 0); return traced; }
        for (; d + 2 <= ta.depth; d += 2) { // render.cpp:140-188
            if (__builtin_expect(step(s, n, d), 0)) { rest_exact(s, d); return traced; }
            if (__builtin_expect(step(n, s, d + 1), 0)) { rest_exact(n, d + 1); return traced; }
        }
        if (d < ta.depth) {
            if (__builtin_expect(step(s, n, d), 0)) { rest_exact(s, d); return traced; }
            s = n;
        }
        s.rxy = rxy; s.rz = rz;
        if (lane_alive) s.alive = select_const(alive, 1);
    }
    return traced;
}

// ---- trace: any scene, LDS-staged tiles -------------------------------------------------
// Every thread of the workgroup must call this together (it contains barriers).
// One segment of every lane of the WORKGROUP against the whole scene (brute force over LDS-staged tiles) followed by the
// shading step; the state of lanes with `fin` is left alone.  Every thread of the workgroup must call this together
// (it contains barriers).  Roulette and counters are the caller's.
template <int MODE>
__device__ __forceinline__ void dyn_segment(const float *__restrict__ sph, float4 *tile, PathState &s, bool fin,
                                            const TraceArgs &ta) {
    const uint32_t ns = ta.ns;
    const float *r2 = sph, *cx = sph + ns, *cy = sph + 2 * (size_t)ns, *cz = sph + 3 * (size_t)ns;
    const float *colx = sph + 7 * (size_t)ns, *coly = sph + 8 * (size_t)ns, *colz = sph + 9 * (size_t)ns;
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
            const HitPre2 h01 = intersect_pre2(a0, c0, s.oxy.x, s.oxy.y, s.oz, s.dxy.x, s.dxy.y, s.dz);
            const HitPre2 h23 = intersect_pre2(a1, c1, s.oxy.x, s.oxy.y, s.oz, s.dxy.x, s.dxy.y, s.dz);
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
    if (!fin) s = n;
}

template <int MODE, bool RETIRE>
__device__ __forceinline__ uint32_t trace_dyn(const float *__restrict__ sph, float4 *tile, PathState &s, bool valid,
                                              const TraceArgs &ta, uint64_t path) {
    const uint64_t rr_key = ta.rr_start ? rr_path_key(ta.seed, path) : 0;
    uint32_t traced = 0;
    for (uint32_t d = 0; d < ta.depth; ++d) {
        const bool fin = !valid || (RETIRE && path_finished(s));
        if (RETIRE && __syncthreads_and(fin)) break;
        dyn_segment<MODE>(sph, tile, s, fin, ta);
        if (!fin) {
            if (ta.rr_start && d + 1 >= ta.rr_start) russian_roulette(s, rr_key, d);
            ++traced;
        }
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
__device__ __forceinline__ void grid_stats(const TraceArgs &ta, uint32_t n_cells, uint32_t n_tests) {
    if (ta.traced) { // statistics: cells visited / candidates tested (per lane, summed over the wave)
        unsigned long long c = n_cells, t = n_tests;
        for (int off = 32; off > 0; off >>= 1) { c += __shfl_xor(c, off, 64); t += __shfl_xor(t, off, 64); }
        if ((threadIdx.x & 63) == 0) { atomicAdd(ta.traced + 1, c); atomicAdd(ta.traced + 2, t); }
    }
}

struct GridCtx { // what one segment through the grid needs besides the path (wave-uniform)
    const GridHeader *h;
    const float *sph;
    const uint32_t *grid;
    const TraceArgs *ta;
};

// One segment (intersect through the grid, shade) for the lanes with !fin; others only take part in the wave-uniform
// parts.  Russian roulette and the counters are the caller's.
template <int MODE>
__device__ __forceinline__ void grid_segment(const GridCtx &ctx, PathState &s, bool fin, uint32_t &n_cells, uint32_t &n_tests) {
    const GridHeader &h = *ctx.h;
    const TraceArgs &ta = *ctx.ta;
    const float *__restrict__ sph = ctx.sph;
    const uint32_t *__restrict__ grid = ctx.grid;
    const uint32_t ns = ta.ns;
    const uint32_t *large = grid + h.off_large, *cells = grid + h.off_cells, *items = grid + h.off_items;
    const float4 *geom = reinterpret_cast<const float4 *>(grid + h.off_geom);
    const float4 *item_geom = reinterpret_cast<const float4 *>(grid + h.off_item_geom);
    const float *colx = sph + 7 * (size_t)ns, *coly = sph + 8 * (size_t)ns, *colz = sph + 9 * (size_t)ns;
    const int n0 = (int)h.n[0], n1 = (int)h.n[1], n2 = (int)h.n[2];
    // A grid built for another scene (or a stale / foreign pointer that still carries the magic) would index past
    // the sphere table: such a buffer is not walked at all -- every sphere is tested straight from the [10][Ns]
    // planes instead (same image, brute-force speed).  Wave-uniform.
    const bool grid_ok = h.magic == kGridMagic && h.num_spheres == ns;
    const float *r2p = sph, *cxp = sph + ns, *cyp = sph + 2 * (size_t)ns, *czp = sph + 3 * (size_t)ns;
        float tmin = kMissT;
    int idx = (MODE == kModeOracle) ? -1 : 0;
    auto test_geom = [&](const float4 g, uint32_t k) {
        const HitPre hp = intersect_pre(g.x, g.y, g.z, g.w, s.oxy.x, s.oxy.y, s.oz, s.dxy.x, s.dxy.y, s.dz);
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
        const HitPre hp = intersect_pre(g.x, g.y, g.z, g.w, s.oxy.x, s.oxy.y, s.oz, s.dxy.x, s.dxy.y, s.dz);
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
        const HitPre hp = intersect_pre(g.x, g.y, g.z, g.w, s.oxy.x, s.oxy.y, s.oz, s.dxy.x, s.dxy.y, s.dz);
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
#if defined(__HIP_DEVICE_COMPILE__)
    if (grid_ok && h.off_cellslot) {
        // The always-tested list from its pair slots, by explicit SCALAR loads (wave-uniform data the compiler would fetch with vector
        // loads and a wait each, like the header: load_grid_header): two spheres and their ids per pair of loads.
        typedef float f32x8 __attribute__((ext_vector_type(8)));
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const float *sg = reinterpret_cast<const float *>(grid + h.off_slots);
        const uint32_t *si = grid + h.off_slot_ids;
        for (uint32_t j = 0; j < h.slot_base; ++j) {
            f32x8 g8;
            u32x2 id2;
            asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx2 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(g8), "=&s"(id2) : "s"(sg + 8 * j), "s"(si + 2 * j) : "memory");
            test_large(make_float4(g8[0], g8[2], g8[4], g8[6]), id2[0]);
            if (id2[1] != kGridNoSphere) test_large(make_float4(g8[1], g8[3], g8[5], g8[7]), id2[1]);
        }
    } else
#endif
    if (grid_ok)
        for (uint32_t i = 0; i < h.nlarge; ++i) { const uint32_t k = large[i]; test_large(geom[k], k); } // wave-uniform
    const float dd = s.dxy.x * s.dxy.x + s.dxy.y * s.dxy.y + s.dz * s.dz;
    const bool unit = fabsf(dd - 1.0f) <= 1e-3f; // false for NaN/inf
    if (!grid_ok) {
        if (!fin)
            for (uint32_t k = 0; k < ns; ++k) { ++n_tests; test_geom(make_float4(cxp[k], cyp[k], czp[k], r2p[k]), k); }
    } else if (!fin && !unit) {
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
        const float ix = recip(s.dxy.x), iy = recip(s.dxy.y), iz = recip(s.dz);
        auto slab = [&](float o, float dv, float inv, float lo, float hi) {
            if (fabsf(dv) > 1e-20f) {
                const float t1 = (lo - o) * inv, t2 = (hi - o) * inv;
                tn = fmaxf(tn, fminf(t1, t2));
                tf = fminf(tf, fmaxf(t1, t2));
            } else if (!(o >= lo && o <= hi)) inbox = false;
        };
        slab(s.oxy.x, s.dxy.x, ix, h.gmin[0], h.gmax[0]);
        slab(s.oxy.y, s.dxy.y, iy, h.gmin[1], h.gmax[1]);
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
            axis(s.oxy.x, s.dxy.x, ix, h.gmin[0], h.cell[0], h.inv_cell[0], n0, c0, st0, tm0, td0);
            axis(s.oxy.y, s.dxy.y, iy, h.gmin[1], h.cell[1], h.inv_cell[1], n1, c1, st1, tm1, td1);
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
        const float4 gc = grid_ok ? geom[g] : make_float4(cxp[g], cyp[g], czp[g], r2p[g]);
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
    }
}

// Can the sample-queue kernel's grid form (pt_queue.h run_grid) render with this grid?  Wave-uniform, read from the buffer
// itself on the device (no host read-back in the launch path): the grid is this scene's, carries the pair-slot tables,
// and eps permits the integer root keys.  The host launches BOTH kernels for such a frame (TraceArgs.grid_walk == 2): the one
// this predicate does not select returns at once.
__device__ __forceinline__ bool grid_queue_usable(const TraceArgs &ta) {
    const GridHeader *h = reinterpret_cast<const GridHeader *>(ta.grid);
    return h && h->magic == kGridMagic && h->num_spheres == ta.ns && h->off_cellslot != 0 && eps_allows_rootkey(ta.eps);
}

// The 128-byte header by two SCALAR loads: it is wave-uniform, but the compiler cannot prove that the kernel's own stores never
// touch the grid buffer and would fetch every field with a vector load (and a wait) at each use.
__device__ __forceinline__ GridHeader load_grid_header(const uint32_t *grid) {
    static_assert(sizeof(GridHeader) == 128, "two s_load_dwordx16");
    GridHeader h;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
    u32x16 w0, w1;
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(w0), "=&s"(w1) : "s"(grid) : "memory");
    uint32_t hw[32];
#pragma unroll
    for (int i = 0; i < 16; ++i) { hw[i] = w0[i]; hw[16 + i] = w1[i]; }
    __builtin_memcpy(&h, hw, sizeof h);
#else
    __builtin_memcpy(&h, grid, sizeof h);
#endif
    return h;
}

template <int MODE, bool RETIRE>
__device__ __forceinline__ uint32_t trace_grid(const float *__restrict__ sph, const uint32_t *__restrict__ grid,
                                               PathState &s, bool valid, const TraceArgs &ta, uint64_t path) {
    const GridHeader hdr = load_grid_header(grid);
    const GridCtx gc{&hdr, sph, grid, &ta};
    const uint64_t rr_key = ta.rr_start ? rr_path_key(ta.seed, path) : 0;
    uint32_t traced = 0, n_cells = 0, n_tests = 0; // statistics
    for (uint32_t d = 0; d < ta.depth; ++d) {
        const bool fin = !valid || (RETIRE && path_finished(s));
        if (RETIRE && __all(fin)) break;
        grid_segment<MODE>(gc, s, fin, n_cells, n_tests);
        if (!fin) {
            if (ta.rr_start && d + 1 >= ta.rr_start) russian_roulette(s, rr_key, d);
            ++traced;
        }
    }
    grid_stats(ta, n_cells, n_tests);
    return traced;
}

// spheres.bin layout [10][8]: r2, x, y, z, em*3, col*3 (gen_data.py:106-127, rt_helper.h:93-102)
// ONES: also write entry 16 = (1,1,1), the "albedo" of a path that is no longer alive (two-path form, pt_trace2.h); the sample-queue kernels
// do not use it and keep their camera there (pt_queue.h: every LDS byte counts towards their occupancy).
template <bool ONES = true>
__device__ __forceinline__ Tab8 load_scene8(const float *__restrict__ sph, Scene8 &sc, float4 *tab) {
#pragma unroll
    for (int k = 0; k < 8; ++k) { // constant offsets from a uniform read-only pointer: scalar loads
        sc.r2[k] = sph[k]; sc.cx[k] = sph[8 + k]; sc.cy[k] = sph[16 + k]; sc.cz[k] = sph[24 + k];
    }
    if (threadIdx.x < 8) {
        const int k = threadIdx.x;
        tab[k] = make_float4(sph[8 + k], sph[16 + k], sph[24 + k], sph[k]);
        tab[8 + k] = make_float4(sph[56 + k], sph[64 + k], sph[72 + k], 0.0f);
    }
    if (ONES && threadIdx.x == 8) tab[16] = make_float4(1.0f, 1.0f, 1.0f, 0.0f); // "albedo" of a path that is no longer alive (pt_trace2.h)
    sc.planes = scene8_shares_planes(sc);
    __syncthreads();
    return Tab8{tab, tab + 8};
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
