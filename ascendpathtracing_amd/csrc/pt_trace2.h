// pt_trace2.h -- TWO paths per lane for the full-trace 8-sphere frame kernel (included by pt_kernels.h).
//
// In this kernel a VALU instruction costs one ~4-cycle issue slot whatever it does (profiles/microbench/
// issue_model_mi355x.txt), so the bounce costs its instruction COUNT and a packed
// v_pk_*_f32 is two operations for one slot.  The intersections are already packed (two spheres per
// instruction); the shading step of ONE path only fills the second half for the x/y components.  With two
// paths A and B in a lane, every register pair holds (A, B) of one quantity and the whole shading step runs
// packed: 36 instead of 48 instructions per path and bounce.  The intersections stay as they are -- eight
// spheres in four packed pairs per path, reading ray component A or B out of its pair through op_sel (free) --
// and run for A, then for B, so the scalar arg-min masks of A are consumed before B needs the registers.
//
// Same arithmetic as bounce_ns8_v2 (pt_trace.h), operation for operation: v_pk_{add,mul,fma}_f32 round each half
// like the scalar instruction.  Used for the frame kernel without APT_FLAG_RETIRE and APT_FLAG_RR only.
#pragma once
#include "pt_trace.h"

namespace {

struct PathPair { // .x = path A, .y = path B
    f2 ox, oy, oz, dx, dy, dz; // rays
    f2 rx, ry, rz;             // throughputs
};

__device__ __forceinline__ PathState unpack_path(const PathPair &p, int which, uint64_t alive) {
    PathState s;
    if (which == 0) path_init(s, p.ox.x, p.oy.x, p.oz.x, p.dx.x, p.dy.x, p.dz.x);
    else path_init(s, p.ox.y, p.oy.y, p.oz.y, p.dx.y, p.dy.y, p.dz.y);
    s.rxy = which == 0 ? f2{p.rx.x, p.ry.x} : f2{p.rx.y, p.ry.y};
    s.rz = which == 0 ? p.rz.x : p.rz.y;
    s.alive = select_const(alive, 1);
    return s;
}

// float32(sum in float64 of three float32 products): np.dot / np.linalg.norm of the NumPy oracle (gen_data.py:347,349).  sdot's start
// value 0.0 is not added: squares are never -0, and the dot product's consumer is twice_canonical() (pt_trace.h).
__device__ __forceinline__ float sum3_f64(float p0, float p1, float p2) {
    double acc = (double)p0;
    acc = acc + (double)p1;
    acc = acc + (double)p2;
    return (float)acc;
}

// One bounce of both paths.  aliveA / aliveB: wave masks (in: before, out: after).  redoA / redoB: wave masks of
// the lanes whose path A / B left the validity range of the fast sequences.  `ones_off`: byte offset, relative to the albedo
// table, of an entry (1, 1, 1) -- a path that is no longer alive multiplies its throughput by it (x1 is exact).
template <int MODE, bool PLANES>
__device__ __forceinline__ void bounce2_ns8(const Scene8 &sc, const Tab8 tab, const PathPair &s, PathPair &n,
                                            const TraceArgs &ta, const KeyConsts &kc, uint32_t ones_off,
                                            uint64_t &aliveA, uint64_t &aliveB, uint64_t &redoA, uint64_t &redoB) {
    float aminA = 1.0f, aminB = 1.0f; // per path: a finished path's request for the exact form can be ignored (trace2_ns8)
    const Hit8 hA = intersect_ns8_v2<MODE, PLANES>(sc, s.ox.x, s.oy.x, s.oz.x, s.dx.x, s.dy.x, s.dz.x, ta, kc, aminA);
    const Hit8 hB = intersect_ns8_v2<MODE, PLANES>(sc, s.ox.y, s.oy.y, s.oz.y, s.dx.y, s.dy.y, s.dz.y, ta, kc, aminB);
    aliveA &= ~hA.light;                    // rt_helper.h:773-787  alive &= idx != light
    aliveB &= ~hB.light;
    // centre and albedo of the two hit spheres: dword reads from the LDS table straight into (A, B) pairs
    const char *geo = reinterpret_cast<const char *>(tab.geo);
    uint32_t cA, cB; // albedo entry, or the (1,1,1) entry once the path is not alive (rt_helper.h:799-810: ret *= alive ? albedo : 1)
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(cA) : "v"(ones_off), "v"(hA.addr), "s"(aliveA));
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(cB) : "v"(ones_off), "v"(hB.addr), "s"(aliveB));
    // Twelve ds_read_b32 in one asm block, each landing in its half of an (A, B) pair.  Left to the compiler the reads of one
    // path are merged into ds_read2_b32 -- an (x, y) pair of ONE path, which then costs six v_mov per pair-bounce (VALU slots) to
    // shuffle into the (A, B) layout; LDS issue is not what binds this kernel.  The block waits for its own reads (the
    // compiler's s_waitcnt insertion does not see them); the other waves of the SIMD cover the latency.
    const uint32_t gA = (uint32_t)(uintptr_t)geo + hA.addr, gB = (uint32_t)(uintptr_t)geo + hB.addr;
    const uint32_t aA = (uint32_t)(uintptr_t)geo + cA, aB = (uint32_t)(uintptr_t)geo + cB; // + 128 below: alb = geo + 8 entries (load_scene8)
    float cxA, cyA, czA, cxB, cyB, czB, axA, ayA, azA, axB, ayB, azB;
    asm volatile("ds_read_b32 %0, %12\n ds_read_b32 %1, %12 offset:4\n ds_read_b32 %2, %12 offset:8\n"
                 "ds_read_b32 %3, %13\n ds_read_b32 %4, %13 offset:4\n ds_read_b32 %5, %13 offset:8\n"
                 "ds_read_b32 %6, %14 offset:128\n ds_read_b32 %7, %14 offset:132\n ds_read_b32 %8, %14 offset:136\n"
                 "ds_read_b32 %9, %15 offset:128\n ds_read_b32 %10, %15 offset:132\n ds_read_b32 %11, %15 offset:136\n"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(cxA), "=&v"(cyA), "=&v"(czA), "=&v"(cxB), "=&v"(cyB), "=&v"(czB), "=&v"(axA), "=&v"(ayA), "=&v"(azA),
                   "=&v"(axB), "=&v"(ayB), "=&v"(azB)
                 : "v"(gA), "v"(gB), "v"(aA), "v"(aB)
                 : "memory");
    const f2 cx = {cxA, cxB}, cy = {cyA, cyB}, cz = {czA, czB};
    const f2 ax = {axA, axB}, ay = {ayA, ayB}, az = {azA, azB};
    // GenerateNewRays, rt_helper.h:504-709 (see bounce_ns8_v2 for the single-path form)
    const f2 t = {hA.tmin, hB.tmin};
    const f2 hx = s.ox + s.dx * t, hy = s.oy + s.dy * t, hz = s.oz + s.dz * t;   // :513-518
    const f2 nx = hx - cx, ny = hy - cy, nz = hz - cz;                            // :635-637
    f2 len2;
    if (MODE == kModeOracle) {
        const f2 p0 = nx * nx, p1 = ny * ny, p2 = nz * nz;
        len2 = f2{sum3_f64(p0.x, p1.x, p2.x), sum3_f64(p0.y, p1.y, p2.y)};
    } else {
        len2 = nx * nx + ny * ny;                                                  // :641-649 (0 + x^2 is x^2)
        len2 = len2 + nz * nz;
    }
    f2 L;
    {   // sqrt_rn_rsq1 on both paths
        const f2 r0 = {__builtin_amdgcn_rsqf(len2.x), __builtin_amdgcn_rsqf(len2.y)};
        aminA = minimum3_abs_after_trans(aminA, r0.x, nx.x);    // validity of the fast sqrt / divide sequences: see kFastMin (pt_trace.h)
        aminA = minimum3_abs(aminA, ny.x, nz.x);
        aminB = minimum3_abs_after_trans(aminB, r0.y, nx.y);
        aminB = minimum3_abs(aminB, ny.y, nz.y);
        const f2 y = len2 * r0, h = r0 * 0.5f;
        const f2 r = __builtin_elementwise_fma(-y, y, len2);
        L = __builtin_elementwise_fma(r, h, y);
    }
    f2 ux, uy, uz;
    {   // div3_packed's sequence (pt_core.h) with (A, B) in the halves: one refined reciprocal per path, three quotients
        const f2 r0 = {__builtin_amdgcn_rcpf(L.x), __builtin_amdgcn_rcpf(L.y)};
        const f2 one = {1.0f, 1.0f};
        const f2 e0 = __builtin_elementwise_fma(-L, r0, one);
        const f2 r = __builtin_elementwise_fma(e0, r0, r0);
        auto quot = [&](const f2 num) __attribute__((always_inline)) {
            f2 q = num * r;
            f2 e = __builtin_elementwise_fma(-L, q, num);
            q = __builtin_elementwise_fma(e, r, q);
            e = __builtin_elementwise_fma(-L, q, num);
            return __builtin_elementwise_fma(e, r, q);
        };
        ux = quot(nx); uy = quot(ny); uz = quot(nz);
    }
    f2 dot;
    if (MODE == kModeOracle) {
        const f2 p0 = s.dx * ux, p1 = s.dy * uy, p2 = s.dz * uz;
        dot = f2{sum3_f64(p0.x, p1.x, p2.x), sum3_f64(p0.y, p1.y, p2.y)};
    } else {
        dot = s.dx * ux;                                                           // :690 Duplicate(0): twice_canonical() below; :694-696
        dot = dot + s.dy * uy;
        dot = dot + s.dz * uz;
    }
    const f2 k2 = twice_canonical(dot);                                            // :697
    n.dx = s.dx - ux * k2; n.dy = s.dy - uy * k2; n.dz = s.dz - uz * k2;           // :699-704
    n.ox = hx; n.oy = hy; n.oz = hz;                                               // :706-708
    n.rx = ax * s.rx; n.ry = ay * s.ry; n.rz = az * s.rz;                          // :804-810 (albedo or 1)
    redoA = __builtin_amdgcn_ballot_w64(!(aminA >= kFastMin)); // something too small, len2 > 2^60, or a NaN
    redoB = __builtin_amdgcn_ballot_w64(!(aminB >= kFastMin));
}

// All bounces of both paths (full trace: no retirement, no roulette).  The hot loop has no merge with the exact
// form and no state copies (two bounces per turn, ping-pong): see trace_ns8.
template <int MODE, bool PLANES>
__device__ __forceinline__ void trace2_ns8_t(const Scene8 &sc, const Tab8 tab, PathPair &s, const TraceArgs &ta) {
    const KeyConsts kc = make_key_consts(ta.eps);
    uint32_t ones_off = 8 * 16; // the entry after the 8 albedos (load_scene8 writes it)
    asm volatile("" : "+v"(ones_off));
    uint64_t aliveA = __builtin_amdgcn_ballot_w64(true), aliveB = aliveA;
    // The exact form of a bounce is a COLD BLOCK INSIDE the loop (the wave redoes that one bounce of both paths
    // with sqrtf() and '/', writes the same registers, and goes on with fast bounces).  The loop then has no exits besides its
    // end -- the round-2 form left the loop for a "rest of the path, exact" tail, and every such exit edge kept the state of
    // both ping-pong halves alive across the back edge.  (Four bounces per turn -- half as many back-edge copies -- measured in
    // round 3: C2 20.40 against 19.95 ms, the longer body costs the ray-generate part more registers than the copies cost; not kept.)
    const bool fast_ok = eps_allows_rootkey(ta.eps);
    auto step = [&](const PathPair &in, PathPair &out) __attribute__((always_inline)) {
        uint64_t oa = aliveA, ob = aliveB;
        bool redo_any = !fast_ok;
        if (__builtin_expect(fast_ok, 1)) {
            uint64_t redoA, redoB;
            bounce2_ns8<MODE, PLANES>(sc, tab, in, out, ta, kc, ones_off, oa, ob, redoA, redoB);
            if (__builtin_expect((redoA | redoB) != 0, 0)) {
                // the request of a path that is already finished (alive bit cleared or throughput zero) is ignored: it cannot
                // reach any output any more (deep all-miss paths, |n| ~ 1e20, are of that kind)
                const bool finA = select_const(aliveA, 1) == 0 || (in.rx.x == 0.0f && in.ry.x == 0.0f && in.rz.x == 0.0f);
                const bool finB = select_const(aliveB, 1) == 0 || (in.rx.y == 0.0f && in.ry.y == 0.0f && in.rz.y == 0.0f);
                redo_any = __builtin_amdgcn_ballot_w64((select_const(redoA, 1) != 0 && !finA) || (select_const(redoB, 1) != 0 && !finB)) != 0;
            }
        }
        if (__builtin_expect(redo_any, 0)) {
            PathState a = unpack_path(in, 0, aliveA), b = unpack_path(in, 1, aliveB), na, nb;
            bounce_ns8_exact<MODE>(sc, tab, a, na, ta);
            bounce_ns8_exact<MODE>(sc, tab, b, nb, ta);
            out.ox = f2{na.oxy.x, nb.oxy.x}; out.oy = f2{na.oxy.y, nb.oxy.y}; out.oz = f2{na.oz, nb.oz};
            out.dx = f2{na.dxy.x, nb.dxy.x}; out.dy = f2{na.dxy.y, nb.dxy.y}; out.dz = f2{na.dz, nb.dz};
            out.rx = f2{na.rxy.x, nb.rxy.x}; out.ry = f2{na.rxy.y, nb.rxy.y}; out.rz = f2{na.rz, nb.rz};
            oa = __builtin_amdgcn_ballot_w64(na.alive != 0);
            ob = __builtin_amdgcn_ballot_w64(nb.alive != 0);
            if (ta.traced && (threadIdx.x & 63) == 0) atomicAdd(ta.traced + 3, 1ull); // statistics: exact re-runs of a wave-bounce
        }
        aliveA = oa; aliveB = ob;
    };
    PathPair n;
    uint32_t d = 0;
    for (; d + 2 <= ta.depth; d += 2) { // render.cpp:140-188
        step(s, n);
        step(n, s);
    }
    if (d < ta.depth) {
        step(s, n);
        s = n;
    }
}

// `planes`: scene8_shares_planes(sc), evaluated once per wave by the kernel (wave-uniform branch around two copies of the loop)
template <int MODE>
__device__ __forceinline__ void trace2_ns8(const Scene8 &sc, const Tab8 tab, PathPair &s, const TraceArgs &ta, bool planes) {
    if (planes) trace2_ns8_t<MODE, true>(sc, tab, s, ta);
    else trace2_ns8_t<MODE, false>(sc, tab, s, ta);
}

} // namespace
