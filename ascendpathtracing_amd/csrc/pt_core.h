// pt_core.h -- arithmetic of the hot path, shared by the gfx950 kernels and the host-side
// helpers of librender_mi355x.so.  Compiled with -ffp-contract=off: every fp32 operation in
// this file is one separately rounded IEEE operation, in the reference's order, because the
// demo scene (r = 1e5 spheres in fp32) amplifies a single differently rounded bit into a
// different path.  fma() appears only where the reference's NumPy
// provably uses one (float64 ddot inside np.linalg.norm).
//
// Reference (paths relative to the reference repository root):
//   intersect_sphere     src/rt_helper.h:255-370  SphereHitInfo
//   running arg-min      src/rt_helper.h:372-451  Transpose + ReduceMinInfo (lowest index on ties)
//   shade_and_reflect    src/rt_helper.h:504-709  GenerateNewRays, :711-830 AccumulateIntervalColor
//   O-mode variants      scripts/gen_data.py:336-349 (np.linalg.norm / np.dot accumulate in float64)
//   camera / tent / ray  scripts/gen_data.py:21-75  gen_rays
#pragma once
#include <stdint.h>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define APT_HD __host__ __device__ __forceinline__
#else
#define APT_HD inline
#endif

namespace apt {

constexpr int kModeKernel = 0;  // APT_MODE_KERNEL
constexpr int kModeOracle = 1;  // APT_MODE_ORACLE
constexpr float kMissT = 1e20f; // rt_helper.h:363, gen_data.py:242

// ---- ray / sphere -------------------------------------------------------------------
// Distance along the ray to sphere (c, r2), or kMissT.  rt_helper.h:263-363.
APT_HD float intersect_sphere(float cx, float cy, float cz, float r2, float ox, float oy, float oz, float dx, float dy,
                              float dz, float eps) {
    float ocx = cx - ox, ocy = cy - oy, ocz = cz - oz;  // :263-268  -(o + (-c)) == c - o exactly
    // :273,:297 the kernel starts b and c from Duplicate(0); 0 + x is exact except that it
    // turns a -0.0 first product into +0.0, which cannot change t (b enters only through
    // b*b and b -/+ q), so the addition is omitted here, as in sim_npu (gen_data.py:206-207).
    float b = ocx * dx;                                 // :278-280  FakeMulAddDst = mul, then add
    b = b + ocy * dy;
    b = b + ocz * dz;
    float c = ocx * ocx;                                // :301-303
    c = c + ocy * ocy;
    c = c + ocz * ocz;
    c = c - r2;                                         // :304
    float disc = b * b;                                 // :314
    disc = disc - c;                                    // :315
    float q = sqrtf(disc);                              // :325 correctly rounded; NaN when disc < 0
    float t0 = b - q, t1 = b + q;                       // :330-331
    float t = (t0 > eps) ? t0 : t1;                     // :341 FakeSelect (NaN compares false)
    return (t > eps) ? t : kMissT;                      // :349,:363
}

// Discriminant only (same operations, same order) for the large-scene traversal, which
// skips the sqrt when no lane of the wave can hit: a miss contributes kMissT, which never
// wins the strict '<' arg-min, so skipping it is result preserving.
struct HitPre { float b, disc; };
APT_HD HitPre intersect_pre(float cx, float cy, float cz, float r2, float ox, float oy, float oz, float dx, float dy,
                            float dz) {
    float ocx = cx - ox, ocy = cy - oy, ocz = cz - oz;
    float b = ocx * dx;
    b = b + ocy * dy;
    b = b + ocz * dz;
    float c = ocx * ocx;
    c = c + ocy * ocy;
    c = c + ocz * ocz;
    c = c - r2;
    float disc = b * b;
    disc = disc - c;
    return {b, disc};
}
APT_HD float intersect_post(HitPre h, float eps) {
    float q = sqrtf(h.disc);
    float t0 = h.b - q, t1 = h.b + q;
    float t = (t0 > eps) ? t0 : t1;
    return (t > eps) ? t : kMissT;
}

#if defined(__HIP_DEVICE_COMPILE__)
// Correctly rounded sqrt for the device hot loop.  hipcc's sqrtf() is v_sqrt_f32 (<= 1 ulp)
// followed by a check of the two neighbouring floats with exact FMA residuals, wrapped in a
// 2^32 pre-scale for inputs below 2^-96 and a class test for 0/inf.  The wrapper is 7 of its
// 16 instructions and is only needed for |x| < 2^-96 (v_sqrt_f32 does not take denormals), so
// the hot loop runs the core alone and reports such inputs through `tiny`; the caller then
// redoes the bounce with sqrtf() (wave-uniform branch, practically never taken; `amin` is the
// running minimum of |x| over the bounce, one v_min_f32 per call).  NaN, +inf
// and negative inputs come out as sqrtf() gives them (NaN compares false in both selects).
// tests/test_gpu_parity.py::test_fast_sqrt_exhaustive checks all 2^32 bit patterns.
__device__ __forceinline__ float sqrt_rn_core(float x, float &amin) {
    amin = fminf(amin, fabsf(x)); // caller tests amin < 2^-96 once per bounce
    float y = __builtin_amdgcn_sqrtf(x);
    const float yd = __int_as_float(__float_as_int(y) - 1);
    const float yu = __int_as_float(__float_as_int(y) + 1);
    const float rd = __builtin_fmaf(-yd, y, x); // x - yd*y, exact sign
    const float ru = __builtin_fmaf(-yu, y, x);
    y = (rd <= 0.0f) ? yd : y;
    y = (ru > 0.0f) ? yu : y;
    return y;
}
// Candidate: Markstein-style correction.  y0 = v_sqrt_f32(x) is within 1 ulp, so the residual
// x - y0*y0 is exactly representable and one FMA with h ~ 1/(2*sqrt(x)) lands on the correctly
// rounded result (sqrt has no exact halfway cases).  5 instructions, no compare/select.
// Verified exhaustively against sqrtf() by the same self-test before it may be used.
__device__ __forceinline__ float sqrt_rn_markstein(float x, float &amin) {
    amin = fminf(amin, fabsf(x));
    const float y = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rsqf(x);
    const float r = __builtin_fmaf(-y, y, x);
    return __builtin_fmaf(r, h, y);
}

// Candidates with ONE transcendental (v_rsq_f32 only), tried by the exhaustive self-test:
//   2: one coupled step          y = x*r, h = r/2, y' = fma(fma(-y,y,x), h, y)
//   3: hipcc's flush-mode lowering (two steps): e = fma(-h,y,.5); y=fma(y,e,y); h=fma(h,e,h); y'=fma(fma(-y,y,x),h,y)
__device__ __forceinline__ float sqrt_rn_rsq1(float x, float &amin) {
    amin = fminf(amin, fabsf(x));
    const float r0 = __builtin_amdgcn_rsqf(x);
    const float y = x * r0, h = 0.5f * r0;
    const float r = __builtin_fmaf(-y, y, x);
    return __builtin_fmaf(r, h, y);
}
__device__ __forceinline__ float sqrt_rn_rsq2(float x, float &amin) {
    amin = fminf(amin, fabsf(x));
    const float r0 = __builtin_amdgcn_rsqf(x);
    float y = x * r0, h = 0.5f * r0;
    const float e = __builtin_fmaf(-h, y, 0.5f);
    y = __builtin_fmaf(y, e, y);
    h = __builtin_fmaf(h, e, h);
    const float r = __builtin_fmaf(-y, y, x);
    return __builtin_fmaf(r, h, y);
}

// Three correctly rounded quotients by one divisor.  hipcc lowers every fp32 `/` to
//   v_div_scale x2, v_rcp, fma, fma (refine 1/d), mul, fma, fma, fma, v_div_fmas, v_div_fixup.
// When neither operand needs scaling (no operand or quotient near the ends of the exponent
// range, numerator non-zero) div_scale and div_fixup are identities and div_fmas is a plain FMA,
// so the reciprocal refinement can be shared by the three numerators: 1 rcp + 2 + 3*5 ops instead
// of 3*11.  The caller redoes the bounce with the plain `/` when the validity flags say so.
// apt_selftest_div3 compares it with `/` on 2^32 structured + random operand sets.
__device__ __forceinline__ void div3_shared(float nx, float ny, float nz, float d, float len2, float &ux, float &uy,
                                            float &uz, float &amin, uint32_t &hiflag) {
    // Validity of the unscaled sequence: numerators not -0 and >= 2^-96 in magnitude (v_div_scale
    // leaves |num| >= 2^-103 alone, and the quotient must stay normal), the divisor d = sqrt(len2)
    // <= 2^30.  The first folds into the sqrt sequences' running minimum `amin` (same threshold
    // 2^-96; +-0 is flagged too, which only costs an exact re-run); the second is the sign bit of
    // bits(2^60) - bits(len2) (len2 >= 0 or NaN), OR-ed into `hiflag` with 2-cycle integer ops.
    // len2 itself must not have underflowed (d would be 0 or unrelated to the numerators).
    // The seed must be v_rcp_f32(d) itself: seeding from the sqrt's v_rsq_f32(len2) (error ~1.5 ulp of
    // 1/d) passes 2^30 random operand sets but fails 0.6 % of the self-test's structured ones (divisor
    // mantissa all ones, where 1/d sits 2^-48 from a rounding midpoint) -- measured, rejected.
    amin = fminf(fminf(amin, fabsf(nx)), fminf(fabsf(ny), fabsf(nz)));
    amin = fminf(amin, len2);
    hiflag |= 0x5d800000u - __float_as_uint(len2);
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e0 = __builtin_fmaf(-d, r0, 1.0f);
    const float r = __builtin_fmaf(e0, r0, r0);
    float q, e;
    q = nx * r; e = __builtin_fmaf(-d, q, nx); q = __builtin_fmaf(e, r, q); e = __builtin_fmaf(-d, q, nx);
    ux = __builtin_fmaf(e, r, q);
    q = ny * r; e = __builtin_fmaf(-d, q, ny); q = __builtin_fmaf(e, r, q); e = __builtin_fmaf(-d, q, ny);
    uy = __builtin_fmaf(e, r, q);
    q = nz * r; e = __builtin_fmaf(-d, q, nz); q = __builtin_fmaf(e, r, q); e = __builtin_fmaf(-d, q, nz);
    uz = __builtin_fmaf(e, r, q);
}


#endif

#if defined(__HIPCC__) // device functions that host-pass code in pt_trace.h / pt_kernels.h names
// The same three quotients with x and y in one register pair (v_pk_mul_f32 / v_pk_fma_f32 round each half exactly
// like the scalar instructions): the form the 8-sphere bounce block uses.  Validity is the caller's business:
// div3_operands_ok() below states it, the bounce block evaluates the same conditions with two v_min3_f32 and two
// compares for the whole wave, and apt_selftest_div3 checks this function under exactly that predicate.
typedef float f2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void div3_packed(f2_t nxy, float nz, float d, f2_t &uxy, float &uz) {
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e0 = __builtin_fmaf(-d, r0, 1.0f);
    const float r = __builtin_fmaf(e0, r0, r0);
    const f2_t nd = {-d, -d}, rr = {r, r};
    f2_t q2 = nxy * r;
    f2_t e2 = __builtin_elementwise_fma(nd, q2, nxy);
    q2 = __builtin_elementwise_fma(e2, rr, q2);
    e2 = __builtin_elementwise_fma(nd, q2, nxy);
    uxy = __builtin_elementwise_fma(e2, rr, q2);
    float q = nz * r, e = __builtin_fmaf(-d, q, nz);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-d, q, nz);
    uz = __builtin_fmaf(e, r, q);
}
// The predicate the 8-sphere bounce block evaluates (pt_trace.h, kFastMin = 2^-29, NaN-propagating minimum): |numerators|
// >= 2^-29 (not -0, no scaling needed, len2 >= 2^-58) and rsq(len2) >= 2^-29, i.e. len2 <= 2^58 up to the rsq's last ulp and
// not NaN -- a subset of what the sequence is valid for (|numerators| and len2 >= 2^-96, len2 <= 2^60: d = sqrt(len2) <= 2^30).
__device__ __forceinline__ bool div3_operands_ok(float len2, float nx, float ny, float nz) {
    const float r0 = __builtin_amdgcn_rsqf(len2);
    const float lo = fminf(fminf(fabsf(nx), fabsf(ny)), fminf(fabsf(nz), r0));
    return lo >= 0x1p-29f && r0 == r0 && nx == nx && ny == ny && nz == nz;
}
#endif

// The two roots b -/+ q of one ray/sphere pair in the reference's own form (rt_helper.h:263-331), sqrtf() for the square root.
APT_HD void intersect_roots(float cx, float cy, float cz, float r2, float ox, float oy, float oz, float dx, float dy,
                            float dz, float &t0, float &t1) {
    float ocx = cx - ox, ocy = cy - oy, ocz = cz - oz;
    float b = ocx * dx;
    b = b + ocy * dy;
    b = b + ocz * dz;
    float c = ocx * ocx;
    c = c + ocy * ocy;
    c = c + ocz * ocz;
    c = c - r2;
    float disc = b * b;
    disc = disc - c;
    const float q = sqrtf(disc);
    t0 = b - q;
    t1 = b + q;
}

// Root selection of rt_helper.h:341-363 as the reference writes it.
APT_HD float select_root(float t0, float t1, float eps) {
    float t = (t0 > eps) ? t0 : t1;
    return (t > eps) ? t : kMissT;
}

// The same selection and the running arg-min in the integer domain, valid for 0 < eps < kMissT.
// For a float x with bit pattern u(x): x is an acceptable root (eps < x, not NaN) exactly when
// key(x) = u(x) - (u(eps)+1), as an unsigned number, is <= u(+inf) - (u(eps)+1); keys of
// acceptable roots are ordered like the roots; negative numbers, zeros, values <= eps and NaNs of
// either sign wrap to keys above every acceptable one.  Since t1 >= t0 whenever both are numbers,
// "t0 if t0 > eps else t1 if t1 > eps else miss" is min(key(t0), key(t1)), and a root beats the
// running minimum (initially kMissT) exactly when its key is below the running key (initially
// key(kMissT)), so tmin = value(running key) needs no select at the end.  (pt_trace.h KeyConsts / intersect_ns8_v2, pt_queue.h test_post.)
APT_HD uint32_t f32_bits(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(x);
#else
    union { float f; uint32_t u; } v; v.f = x; return v.u;
#endif
}
APT_HD float bits_f32(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    union { float f; uint32_t u; } v; v.u = u; return v.f;
#endif
}
APT_HD bool eps_allows_rootkey(float eps) { return eps > 0.0f && eps < kMissT; }

#if defined(__clang__) // everything from here to the matching #endif is used by the kernels only (hipcc = clang)
// State of one path between bounces.  The x and y components of the three 3-vectors are kept as register PAIRS
// (ext_vector float2): the bounce block runs them through packed v_pk_*_f32 instructions, and a pair built on
// the fly from two scalars would cost a register copy per bounce.
typedef float f2 __attribute__((ext_vector_type(2)));
struct PathState {
    f2 oxy, dxy;                  // ray (updated in place, rt_helper.h:699-708)
    f2 rxy;                       // throughput `ret` (render.cpp:116-121)
    float oz, dz, rz;
    uint32_t alive;               // retMask bit (render.cpp:123-124); a dword so the struct has no padding
};

APT_HD void path_init(PathState &s, float ox, float oy, float oz, float dx, float dy, float dz) {
    s.oxy = f2{ox, oy}; s.oz = oz; s.dxy = f2{dx, dy}; s.dz = dz;
    s.rxy = f2{1.0f, 1.0f}; s.rz = 1.0f;
    s.alive = 1u;
}

// A finished path: nothing a further bounce does can change its colour (Appendix A notes).
APT_HD bool path_finished(const PathState &s) {
    return !s.alive || (s.rxy.x == 0.0f && s.rxy.y == 0.0f && s.rz == 0.0f);
}

// GenerateNewRays + AccumulateIntervalColor for the hit (tmin, sphere centre c, albedo col).
// is_light: the arg-min index equals light_index.
template <int MODE, bool FAST = false>
APT_HD void shade_and_reflect(PathState &s, float tmin, float cx, float cy, float cz, float colx, float coly,
                              float colz, bool is_light, float *amin = nullptr) {
    float hx = s.dxy.x * tmin, hy = s.dxy.y * tmin, hz = s.dz * tmin; // rt_helper.h:513-518
    hx = s.oxy.x + hx; hy = s.oxy.y + hy; hz = s.oz + hz;
    float nx = hx - cx, ny = hy - cy, nz = hz - cz;             // :635-637
    float L;
    if (MODE == kModeOracle) {                                  // np.linalg.norm, gen_data.py:347
        float p0 = nx * nx, p1 = ny * ny, p2 = nz * nz;
        double acc = 0.0 + (double)p0;                          // sdot: double dot = 0.0; dot += y*x
        acc = acc + (double)p1;
        acc = acc + (double)p2;
        L = (float)acc;
    } else {
        float acc = 0.0f + nx * nx;                             // :641 Duplicate(0), :647-649
        acc = acc + ny * ny;
        acc = acc + nz * nz;
        L = acc;
    }
    const float len2 = L; (void)len2;
#if defined(__HIP_DEVICE_COMPILE__)
    if (FAST) L = sqrt_rn_rsq1(L, *amin);   // exact for every float but +inf, which div3_shared's hiflag sends to the re-run
    else
#endif
        L = sqrtf(L);                                           // :658
    float ux, uy, uz;                                           // :664-666 IEEE divide
#if defined(__HIP_DEVICE_COMPILE__)
    if (FAST) {
        uint32_t hiflag = 0;
        div3_shared(nx, ny, nz, L, len2, ux, uy, uz, *amin, hiflag);
        if ((int32_t)hiflag < 0) *amin = 0.0f;                  // forces the exact re-run of this bounce
    } else
#endif
    {
        ux = nx / L; uy = ny / L; uz = nz / L;
    }
    float dot;
    if (MODE == kModeOracle) {                                  // np.dot, gen_data.py:349
        float p0 = s.dxy.x * ux, p1 = s.dxy.y * uy, p2 = s.dz * uz;
        double acc = 0.0 + (double)p0;
        acc = acc + (double)p1;
        acc = acc + (double)p2;
        dot = (float)acc;
    } else {
        dot = 0.0f + s.dxy.x * ux;                                 // :690 Duplicate(0), :694-696
        dot = dot + s.dxy.y * uy;
        dot = dot + s.dz * uz;
    }
    float k2 = dot * 2.0f;                                      // :697
    float mx = ux * k2, my = uy * k2, mz = uz * k2;             // :699-701
    s.dxy.x = s.dxy.x - mx; s.dxy.y = s.dxy.y - my; s.dz = s.dz - mz;       // :702-704
    s.oxy.x = hx; s.oxy.y = hy; s.oz = hz;                            // :706-708
    s.alive = (s.alive && !is_light) ? 1u : 0u;                 // :773-787
    if (s.alive) {                                              // :799-810 (x1 is exact otherwise)
        s.rxy.x = colx * s.rxy.x; s.rxy.y = coly * s.rxy.y; s.rz = colz * s.rz;
    }
}

#endif // __clang__

// ---- first-hit debug mode: scripts/gen_data.py:134-188 test_scene --------------------------
// One ray against sphere k with test_scene's arithmetic: np.dot on float32 3-vectors (float64
// accumulation of float32 products, rounded once), det = (b*b - dot(op,op)) + r2, miss when
// det < 0.  Updates (mind, id) with the reference's t0-then-t1 rule (:163-168).
APT_HD void test_scene_sphere(float cx, float cy, float cz, float r2, float ox, float oy, float oz, float dx, float dy,
                              float dz, float eps, int k, float &mind, int &id) {
    const float opx = cx - ox, opy = cy - oy, opz = cz - oz;              // :151
    const float p0 = opx * dx, p1 = opy * dy, p2 = opz * dz;
    double acc = 0.0 + (double)p0; acc = acc + (double)p1; acc = acc + (double)p2;
    const float b = (float)acc;                                           // :153
    const float q0 = opx * opx, q1 = opy * opy, q2 = opz * opz;
    acc = 0.0 + (double)q0; acc = acc + (double)q1; acc = acc + (double)q2;
    float det = b * b - (float)acc;                                       // :154
    det = det + r2;
    if (det < 0.0f) return;                                               // :155 (NaN falls through like NumPy)
    det = sqrtf(det);
    const float t0 = b - det, t1 = b + det;
    if (t0 > eps && t0 < mind) { mind = t0; id = k; }                     // :163-165
    else if (t1 > eps && t1 < mind) { mind = t1; id = k; }                // :166-168
}

// ---- ray generation (all float64, cast to float32 at the end: gen_data.py:71) ---------
struct Camera { double pos[3], g[3], cx[3], cy[3], inv_w, inv_h; }; // inv_*: host-made RN(1/w), RN(1/h)

APT_HD double norm3_sq(double x, double y, double z) { // np.linalg.norm: sqrt(ddot); ddot is an FMA chain
    double acc = x * x;
    acc = fma(y, y, acc);
    return fma(z, z, acc);
}
APT_HD double norm3(double x, double y, double z) { return sqrt(norm3_sq(x, y, z)); }

inline void camera_init(Camera &c, uint32_t w, uint32_t h) { // gen_data.py:24-30
    c.pos[0] = 50; c.pos[1] = 52; c.pos[2] = 295.6;
    const double dir[3] = {0, -0.042612, -1};
    const double n = norm3(dir[0], dir[1], dir[2]);
    for (int i = 0; i < 3; ++i) c.g[i] = dir[i] / n;
    c.cx[0] = (double)w * 0.5135 / (double)h; c.cx[1] = 0; c.cx[2] = 0;
    double cr[3];                                             // np.cross(cx, g)
    cr[0] = c.cx[1] * c.g[2] - c.cx[2] * c.g[1];
    cr[1] = c.cx[2] * c.g[0] - c.cx[0] * c.g[2];
    cr[2] = c.cx[0] * c.g[1] - c.cx[1] * c.g[0];
    const double cn = norm3(cr[0], cr[1], cr[2]);
    for (int i = 0; i < 3; ++i) c.cy[i] = cr[i] / cn * 0.5135;
    // camera_ray relies on these (they follow from dir = (0, ., .) and cx = (., 0, 0))
    if (!(c.cx[1] == 0 && c.cx[2] == 0 && c.cy[0] == 0 && !std::signbit(c.cy[0]) && c.g[0] == 0 && !std::signbit(c.g[0]) &&
          c.g[1] != 0 && c.g[2] != 0 && c.cx[0] > 0)) abort();
    c.inv_w = 1.0 / (double)w; // correctly rounded (IEEE division on the host); used by the device's fast quotients
    c.inv_h = 1.0 / (double)h;
}

// What camera_ray() reads of a Camera, and nothing else (80 bytes): the sample-queue kernels keep this form in LDS, where every byte counts
// towards the occupancy (pt_queue.h).  camera_ray_t() reads either type through the accessors below.
struct CameraLite { double cx0, cy1, cy2, g1, g2, pos[3], inv_w, inv_h; };
APT_HD CameraLite camera_lite(const Camera &c) { return CameraLite{c.cx[0], c.cy[1], c.cy[2], c.g[1], c.g[2], {c.pos[0], c.pos[1], c.pos[2]}, c.inv_w, c.inv_h}; }
APT_HD double cam_cx0(const Camera &c) { return c.cx[0]; }
APT_HD double cam_cy1(const Camera &c) { return c.cy[1]; }
APT_HD double cam_cy2(const Camera &c) { return c.cy[2]; }
APT_HD double cam_g1(const Camera &c) { return c.g[1]; }
APT_HD double cam_g2(const Camera &c) { return c.g[2]; }
APT_HD double cam_cx0(const CameraLite &c) { return c.cx0; }
APT_HD double cam_cy1(const CameraLite &c) { return c.cy1; }
APT_HD double cam_cy2(const CameraLite &c) { return c.cy2; }
APT_HD double cam_g1(const CameraLite &c) { return c.g1; }
APT_HD double cam_g2(const CameraLite &c) { return c.g2; }

#if defined(__HIP_DEVICE_COMPILE__)
// float64 sqrt for ray-generate: the core of hipcc's own expansion (v_rsq_f64, coupled Goldschmidt
// steps, two residual corrections) without its ldexp pre/post-scaling (only needed below 2^-767) and
// class test (zero/inf).  Arguments here lie in {0} u [2^-53, 2]; zero or anything below 2^-60 sends
// the wave to sqrt() (camera_ray_t's combined test).  Same operations in the same order as the compiler's sequence, so the same bits
// (checked against the CPU's correctly rounded sqrt by the ray tests, incl. 530 M paths of C2).
__device__ __forceinline__ double sqrt_f64_core(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    double d = fma(-g, g, x);
    h = fma(h, r, h);
    g = fma(d, h, g);
    d = fma(-g, g, x);
    return fma(d, h, g);
}
#endif

// gen_data.py:37-40: dx = sqrt(r)-1 if r < 1 else 1-sqrt(2-r).  Written branch-free (one sqrt of the
// selected argument, then the selected combination): the same operations on the same values as the
// two-armed form, without executing both arms when the lanes of a wave disagree.
// FAST (device only): the exact fast sqrt core, valid for arguments >= 2^-60; the argument is handed back through
// `arg` for camera_ray_t's combined validity test.
template <bool FAST>
APT_HD double tent_t(double u, double &arg) {
    const double r = 2 * u;
    const bool lower = r < 1;
    const double x = lower ? r : 2 - r;
    arg = x;
    double q;
#if defined(__HIP_DEVICE_COMPILE__)
    if (FAST) q = sqrt_f64_core(x);
    else
#endif
        q = sqrt(x);
    // 1 - q is -(q - 1) bit for bit except for q == 1 (u == 0.5 exactly), where it is +0 instead of -0: the only
    // consumer adds the result to sx + 0.5 >= 0.5, which either zero leaves unchanged.
    const double t = q - 1;
    return lower ? t : -t;
}

struct Ray { float ox, oy, oz, dx, dy, dz; };

#if defined(__HIP_DEVICE_COMPILE__)
// Correctly rounded float64 quotients without the full division expansion (v_div_scale x2, v_rcp_f64,
// two Newton steps, mul, fma, v_div_fmas, v_div_fixup: ~56 cycles each, five per ray).
//   div_by_rn_reciprocal: y = RN(1/b) exactly (made on the host) -> q = a*y, r = fma(-b,q,a), q' = fma(r,y,q)
//   is RN(a/b) (Markstein): the same final correction step hipcc's own lowering ends with.
//   refined_reciprocal: v_rcp_f64 + two Newton steps = the reciprocal hipcc's lowering uses; one such
//   reciprocal serves the three quotients by the norm.
// Valid while nothing is scaled: operands and quotients far from the ends of the exponent range and the
// numerator not -0; otherwise the whole wave redoes the quotients with '/'.  Validated bit for bit against true IEEE division on the CPU
// by the device-vs-oracle ray tests, incl. all 530,841,600 paths of config C2.
__device__ __forceinline__ double div_by_rn_reciprocal(double a, double b, double y) {
    const double q = a * y;
    const double r = fma(-b, q, a);
    return fma(r, y, q);
}
__device__ __forceinline__ double min3_abs_f64(double a, double b, double c) { // min(a, |b|, |c|) in two instructions, no canonicalising copies
    double r;
    asm("v_min_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b));
    asm("v_min_f64 %0, %1, |%2|" : "=v"(r) : "v"(r), "v"(c));
    return r;
}
__device__ __forceinline__ double refined_reciprocal(double b) {
    double y = __builtin_amdgcn_rcp(b);
    double e = fma(-b, y, 1.0);
    y = fma(y, e, y);
    e = fma(-b, y, 1.0);
    return fma(y, e, y);
}
// Validity of the fast sequences is checked ONCE per ray: `lo` = min over the two tent arguments, |xa|, |xb|,
// |d0|, |d1|, |d2| and |d|^2 must be >= 2^-60 (nothing is zero -- its sign would matter --, denormal or absurdly
// small: a numerator is 0 only for a handful of exact jitter values, probability ~2^-52 per path) and the norm
// <= 2^60 (also rejects NaN / inf); everything else is bounded by construction (|x| <= w + 1, |d| < 3 with u in
// [0, 1) and a finite camera).  If any lane of the wave fails, the whole wave recomputes the ray with sqrt() and
// '/'.  (One combined test instead of five wave-level branches with their own compare / select / compare
// sequences: 17 instructions per ray less.)
#endif

// Outputs are six scalar references on purpose: an aggregate result gets its stores merged into
// vector stores to a stack slot that SROA can then no longer promote (it ended up in scratch).
template <bool FAST, class CAM>
APT_HD bool camera_ray_t(const CAM &c, uint32_t w, uint32_t h, uint32_t i, uint32_t j, uint32_t sy, uint32_t sx,
                         double u1, double u2, float &rox, float &roy, float &roz, float &rdx, float &rdy, float &rdz) {
    double arg1, arg2;
    const double ddx = tent_t<FAST>(u1, arg1), ddy = tent_t<FAST>(u2, arg2);
    const double xa = ((double)sx + 0.5 + ddx) / 2 + (double)i, xb = ((double)sy + 0.5 + ddy) / 2 + (double)j;
    double a, b;
#if defined(__HIP_DEVICE_COMPILE__)
    if (FAST) {
        a = div_by_rn_reciprocal(xa, (double)w, c.inv_w) - 0.5;                  // :41
        b = div_by_rn_reciprocal(xb, (double)h, c.inv_h) - 0.5;                  // :42
    } else
#endif
    {
        a = xa / (double)w - 0.5;                                                // :41
        b = xb / (double)h - 0.5;                                                // :42
    }
    // :41-43 d = cx*a + cy*b + g with the reference's camera frame, in which cx = (cx0, 0, 0), cy = (+0, cy1, cy2)
    // and g = (+0, g1, g2) (camera_init checks it).  The vanishing terms are exact identities, signs of zero
    // included: a = q - 0.5 is never -0, so cx0*a + (+-0) + (+0) is cx0*a bit for bit; (+-0 + cy1*b) + g1 with
    // g1 != 0 is cy1*b + g1.  Seven float64 operations per ray less than the general form the oracle keeps.
    const double d0 = cam_cx0(c) * a;
    const double d1 = cam_cy1(c) * b + cam_g1(c);
    const double d2 = cam_cy2(c) * b + cam_g2(c);
    const double n2 = norm3_sq(d0, d1, d2);
    rox = (float)(c.pos[0] + d0 * 140);                                      // :45
    roy = (float)(c.pos[1] + d1 * 140);
    roz = (float)(c.pos[2] + d2 * 140);
#if defined(__HIP_DEVICE_COMPILE__)
    if (FAST) {
        const double n = sqrt_f64_core(n2);
        const double y = refined_reciprocal(n);
        rdx = (float)div_by_rn_reciprocal(d0, n, y);                         // :46
        rdy = (float)div_by_rn_reciprocal(d1, n, y);
        rdz = (float)div_by_rn_reciprocal(d2, n, y);
        double lo = min3_abs_f64(arg1, arg2, xa); // everything the fast sequences need bounded away from zero
        lo = min3_abs_f64(lo, xb, d0);
        lo = min3_abs_f64(lo, d1, d2);
        asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(lo), "v"(n2));
        return lo >= 0x1p-60 && n <= 0x1p60;
    }
#endif
    const double n = sqrt(n2);
    rdx = (float)(d0 / n);                                                   // :46
    rdy = (float)(d1 / n);
    rdz = (float)(d2 / n);
    return true;
}

// Outputs are six scalar references on purpose: an aggregate result gets its stores merged into
// vector stores to a stack slot that SROA can then no longer promote (it ended up in scratch).
template <class CAM>
APT_HD void camera_ray(const CAM &c, uint32_t w, uint32_t h, uint32_t i, uint32_t j, uint32_t sy, uint32_t sx,
                       double u1, double u2, float &rox, float &roy, float &roz, float &rdx, float &rdy, float &rdz) {
#if defined(__HIP_DEVICE_COMPILE__)
    const bool ok = camera_ray_t<true>(c, w, h, i, j, sy, sx, u1, u2, rox, roy, roz, rdx, rdy, rdz);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok) == 0, 1)) return;
    asm volatile("" ::: "memory"); // keeps the exact form out of the hot path's schedule
#endif
    (void)camera_ray_t<false>(c, w, h, i, j, sy, sx, u1, u2, rox, roy, roz, rdx, rdy, rdz);
}
APT_HD Ray camera_ray(const Camera &c, uint32_t w, uint32_t h, uint32_t i, uint32_t j, uint32_t sy, uint32_t sx,
                      double u1, double u2) {
    Ray r;
    camera_ray(c, w, h, i, j, sy, sx, u1, u2, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz);
    return r;
}

// Counter-based generator for on-device ray generation (path_uniforms below); xorshift64* serves the
// build-defined scene generator.  Integer-only, so host and device agree bit for bit.
APT_HD uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
APT_HD uint64_t xorshift64s(uint64_t &s) {
    uint64_t x = s;
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    s = x;
    return x * 0x2545F4914F6CDD1Dull;
}
// 52 high bits of z as the mantissa of a double in [1, 2), minus 1: a uniform multiple of 2^-52 in [0, 1) with
// two integer operations and one exact subtraction (a u64 -> f64 conversion costs six float64-class operations).
APT_HD double unit_from_bits(uint64_t z) {
    const uint64_t b = (z >> 12) | 0x3FF0000000000000ull;
#if defined(__HIP_DEVICE_COMPILE__)
    return __longlong_as_double((long long)b) - 1.0;
#else
    double d;
    memcpy(&d, &b, sizeof d);
    return d - 1.0;
#endif
}
// The two uniforms of path p are outputs 2p+1 and 2p+2 of ONE SplitMix64 generator (state += phi; output =
// mix(state)) whose state starts at splitmix64(seed): every path reads its own two consecutive outputs of the
// same well-tested stream by random access, state(p) = splitmix64(seed) + 2p*phi.  52 high bits each.
constexpr uint64_t kPathStride = 0x3C6EF372FE94F82Aull; // 2*phi mod 2^64: the generator state advances by this per path
APT_HD void path_uniforms_at(uint64_t state, double &u1, double &u2) {   // state = splitmix64(seed) + path * kPathStride
    u1 = unit_from_bits(splitmix64(state));
    u2 = unit_from_bits(splitmix64(state + 0x9E3779B97F4A7C15ull));
}
APT_HD void path_uniforms(uint64_t seed, uint64_t path, double &u1, double &u2) {
    path_uniforms_at(splitmix64(seed) + path * kPathStride, u1, u2);
}

// ---- uniform grid over the small spheres of a large scene ----------------------------------------
// One flat buffer of 32-bit words, built on the host (host_helpers.cpp) or on the device (pt_grid_build.h), traversed on the device:
//   GridHeader | large[nlarge] | cell_start[ncells+1] | items[nitems] | geom[Ns] float4 (cx,cy,cz,r2)
//              | item_geom[nitems] float4 = geom[items[i]]  (cell order: the walk reads it directly, one
//                dependent load fewer per candidate)
//              | cellslot[(n0+2)(n1+2)(n2+2)] | slot_geom[nslots] (8 words) | slot_ids[nslots] (2 words)  -- the PAIR-SLOT tables of the flat walk
//              | sphere8[Ns] (8 words)
// Pair slots (round 3; pt_trace.h grid_segment_flat): the candidates of a list, two per slot, laid out for the packed
// two-spheres-per-instruction test and for ONE address per pair:
//   slot_geom = (cx_a, cx_b, cy_a, cy_b | cz_a, cz_b, r2_a, r2_b), slot_ids = (id_a, id_b)      an odd list ends with a NaN sphere, id 0xffffffff
// Slots [0, slot_base) hold the always-tested large list, then every cell's list from slot
//   grid_slot_begin(h, cell_start[c], c) = slot_base + ((cell_start[c] + c + 1) >> 1)
// (lists of ceil(k / 2) slots never overlap under this rule -- no second scan; the few gap slots stay zero and are never read).
// cellslot is indexed by BORDERED cell coordinates, ((z + 1) * (n1 + 2) + (y + 1)) * (n0 + 2) + (x + 1): one layer of cells around the grid holds
// kGridCellOutside, so that a walk that leaves the box reads "outside" where it reads the next cell's range and keeps no count of the steps it has
// left (round 4: six vector instructions per loop turn of the walk).  An inner entry = slot_begin << 7 | slots of the cell, or | 63 for "62 or more:
// take the count from cell_start" (62 in the count field is the outside mark).
// off_cellslot == 0: the tables are absent (they would not fit 26 bits of slot index) and the walk uses item_geom.
struct GridHeader {
    uint32_t magic, num_spheres;
    uint32_t n[3], ncells, nlarge, nitems;
    uint32_t off_large, off_cells, off_items, off_geom;   // word offsets from the start of the buffer
    uint32_t off_item_geom;
    float gmin[3], gmax[3], cell[3], inv_cell[3];
    float margin;                                         // how far every small sphere's box was inflated
    uint32_t off_cellslot, off_slots, off_slot_ids, nslots, slot_base;  // pair-slot tables (0 = absent)
    uint32_t off_sphere8;                                 // [Ns] x 8 words (cx, cy, cz, r2, albedo r, g, b, 0): what the shading step gathers, in ONE cache line
};
constexpr uint32_t kGridSlotCountBits = 6;       // cellslot: low bits = slots of the cell, saturating
constexpr uint32_t kGridSlotCountMax = (1u << kGridSlotCountBits) - 1u;
constexpr uint32_t kGridSlotShift = kGridSlotCountBits + 1u;   // cellslot: slot_begin << 7, i.e. (entry >> 6) IS the list's first POSITION (2 per slot): one shift in the walk
constexpr uint32_t kGridCellOutside = kGridSlotCountMax - 1u;   // cellslot entry of the border layer (count field 62, slot 0)
APT_HD uint32_t grid_bordered_cells(const uint32_t n[3]) { return (n[0] + 2u) * (n[1] + 2u) * (n[2] + 2u); }
constexpr uint32_t kGridNoSphere = 0xffffffffu;  // id of a pad
constexpr uint32_t kGridMagic = 0x47524944u; // "GRID"
constexpr double kGridSpheresPerCell = 0.5;     // default cell size of both builders: sphere centres per cell (APT_GRID_SPHERES_PER_CELL overrides)
constexpr uint32_t kGridMaxCellsPerAxis = 512;  // round 1 capped the grid at 128 cells per axis

// The header of the grid from the statistics of the scene's small spheres (host arithmetic, shared by apt_build_grid_host
// and apt_build_grid_device so that both produce the same bytes): lo/hi = bounding box of the small spheres' own boxes,
// scale = max(|centre| + radius, 1).  Cells are sized for ~per_cell sphere centres each.  Offsets that depend on the
// item count are filled in by grid_header_offsets() once it is known.
inline void grid_header_from_stats(uint32_t ns, uint32_t nsmall, uint32_t nlarge, const float lo_in[3], const float hi_in[3],
                                   float scale, double per_cell, GridHeader &h) {
    memset(&h, 0, sizeof h);
    h.magic = kGridMagic; h.num_spheres = ns; h.nlarge = nlarge;
    float lo[3] = {0, 0, 0}, hi[3] = {1, 1, 1};
    if (nsmall) for (int a = 0; a < 3; ++a) { lo[a] = lo_in[a]; hi[a] = hi_in[a]; }
    else scale = 1.0f;
    h.margin = 0.05f + 1e-4f * scale;
    for (int a = 0; a < 3; ++a) { lo[a] -= 2 * h.margin; hi[a] += 2 * h.margin; }
    const double vol = (double)(hi[0] - lo[0]) * (hi[1] - lo[1]) * (hi[2] - lo[2]);
    const double target = fmax(1.0, (double)nsmall / per_cell);
    const double edge = cbrt(fmax(vol, 1e-30) / target);
    for (int a = 0; a < 3; ++a) {
        const double n = ceil((hi[a] - lo[a]) / fmax(edge, 1e-30));
        h.n[a] = (uint32_t)fmin((double)kGridMaxCellsPerAxis, fmax(1.0, n));
        h.gmin[a] = lo[a]; h.gmax[a] = hi[a];
        h.cell[a] = (hi[a] - lo[a]) / (float)h.n[a];
        h.inv_cell[a] = 1.0f / h.cell[a];
    }
    h.ncells = h.n[0] * h.n[1] * h.n[2];
}
inline size_t grid_header_offsets(GridHeader &h, uint32_t nitems) { // -> total words
    h.nitems = nitems;
    h.off_large = (uint32_t)(sizeof(GridHeader) / 4);
    h.off_cells = h.off_large + h.nlarge;
    h.off_items = h.off_cells + h.ncells + 1;
    h.off_geom = (h.off_items + h.nitems + 3u) & ~3u;                               // 16-byte aligned float4s
    h.off_item_geom = h.off_geom + 4 * h.num_spheres;
    size_t words = (size_t)h.off_item_geom + 4 * (size_t)h.nitems;
    // pair-slot tables: capacity from the placement rule (grid_slot_begin): the last list ends at or before this slot
    h.slot_base = (h.nlarge + 1u) >> 1;
    const uint64_t nslots = (uint64_t)h.slot_base + (((uint64_t)nitems + h.ncells + 1u) >> 1) + 1u;
    const uint64_t off_cellslot = words, off_slots = (off_cellslot + grid_bordered_cells(h.n) + 7u) & ~(uint64_t)7u;   // 32-byte aligned slots
    const uint64_t off_ids = off_slots + nslots * 8u, off_s8 = (off_ids + nslots * 2u + 7u) & ~(uint64_t)7u, end = off_s8 + 8ull * h.num_spheres;
    if (nslots < (1ull << (32 - kGridSlotShift)) && end < (1ull << 32)) {
        h.off_cellslot = (uint32_t)off_cellslot; h.off_slots = (uint32_t)off_slots; h.off_slot_ids = (uint32_t)off_ids; h.nslots = (uint32_t)nslots;
        h.off_sphere8 = (uint32_t)off_s8;
        words = (size_t)end;
    } else {
        h.off_cellslot = h.off_slots = h.off_slot_ids = h.nslots = h.off_sphere8 = 0;
    }
    return words;
}
APT_HD uint32_t grid_slot_begin(const GridHeader &h, uint32_t cell_start_c, uint32_t c) { return h.slot_base + ((cell_start_c + c + 1u) >> 1); }
APT_HD uint32_t grid_cellslot_entry(const GridHeader &h, uint32_t b, uint32_t e, uint32_t c) {
    const uint32_t n = (e - b + 1u) >> 1;
    return grid_slot_begin(h, b, c) << kGridSlotShift | (n < kGridCellOutside ? n : kGridSlotCountMax);
}
// The slots of one sorted id list (n ids from `ids`), starting at slot `slot`; geometry from the geom[] table already in the buffer.
APT_HD void grid_fill_slots(uint32_t *w, const GridHeader &h, uint32_t slot, const uint32_t *ids, uint32_t n) {
    for (uint32_t i = 0; i < n; i += 2, ++slot) {
        uint32_t *g = w + h.off_slots + 8 * (size_t)slot, *id = w + h.off_slot_ids + 2 * (size_t)slot;
        const uint32_t ka = ids[i], kb = i + 1 < n ? ids[i + 1] : kGridNoSphere;
        const uint32_t *ga = w + h.off_geom + 4 * (size_t)ka;
        g[0] = ga[0]; g[2] = ga[1]; g[4] = ga[2]; g[6] = ga[3];
        if (kb != kGridNoSphere) { const uint32_t *gb = w + h.off_geom + 4 * (size_t)kb; g[1] = gb[0]; g[3] = gb[1]; g[5] = gb[2]; g[7] = gb[3]; }
        else g[1] = g[3] = g[5] = g[7] = 0x7fc00000u;   // a NaN sphere: its discriminant is NaN, it can never be hit
        id[0] = ka; id[1] = kb;
    }
}
// sphere8 record of sphere k from the [10][Ns] table
APT_HD void grid_fill_sphere8(uint32_t *w, const GridHeader &h, const float *sph, uint32_t k) {
    const size_t ns = h.num_spheres;
    float *r = reinterpret_cast<float *>(w + h.off_sphere8) + 8 * (size_t)k;
    r[0] = sph[ns + k]; r[1] = sph[2 * ns + k]; r[2] = sph[3 * ns + k]; r[3] = sph[k];
    r[4] = sph[7 * ns + k]; r[5] = sph[8 * ns + k]; r[6] = sph[9 * ns + k]; r[7] = 0.0f;
}
// cellslot entry t of the BORDERED table and the slots of its cell (t == grid_bordered_cells: the always-tested list); needs cell_start, items,
// large and geom in place
APT_HD void grid_fill_cell_slots(uint32_t *w, const GridHeader &h, uint32_t t) {
    if (t == grid_bordered_cells(h.n)) { grid_fill_slots(w, h, 0, w + h.off_large, h.nlarge); return; }
    const uint32_t sx = h.n[0] + 2u, sy = h.n[1] + 2u;
    const uint32_t xb = t % sx, yb = (t / sx) % sy, zb = t / (sx * sy);
    if (xb == 0 || yb == 0 || zb == 0 || xb == h.n[0] + 1u || yb == h.n[1] + 1u || zb == h.n[2] + 1u) { w[h.off_cellslot + t] = kGridCellOutside; return; }
    const uint32_t c = ((zb - 1u) * h.n[1] + (yb - 1u)) * h.n[0] + (xb - 1u);
    const uint32_t b = w[h.off_cells + c], e = w[h.off_cells + c + 1];
    w[h.off_cellslot + t] = grid_cellslot_entry(h, b, e, c);
    grid_fill_slots(w, h, grid_slot_begin(h, b, c), w + h.off_items + b, e - b);
}
// cells [c0, c1] of axis a that the box of a small sphere (centre c, radius rad), inflated by the margin, touches
APT_HD void grid_cell_range(const GridHeader &h, float c, float rad, int a, uint32_t &c0, uint32_t &c1) {
    const float a0 = (c - rad - h.margin - h.gmin[a]) * h.inv_cell[a], a1 = (c + rad + h.margin - h.gmin[a]) * h.inv_cell[a];
    const double hi = (double)h.n[a] - 1.0;
    const double f0 = floor((double)a0), f1 = floor((double)a1);
    c0 = (uint32_t)(f0 < 0.0 ? 0.0 : (f0 > hi ? hi : f0));   // NaN never reaches here: non-finite spheres are "large"
    c1 = (uint32_t)(f1 < 0.0 ? 0.0 : (f1 > hi ? hi : f1));
}
// Does the small sphere (centre, rad), inflated by the margin, touch cell (x, y, z)?  Squared distance from the centre to the cell's box against
// (rad + margin)^2, with a relative slack so that rounding can only KEEP a cell.  The box test of grid_cell_range lists a sphere in the corner
// cells of its bounding box too; a ray can only be accepted by the sphere at a point within rad (1 + fp error) of the centre, and that point
// lies in a cell whose box is at most that far from the centre, so corner cells beyond rad + margin can never matter (same argument, same
// margin, as for the box).  Same operations on host and device (no contraction): both builders make the same lists.
APT_HD bool grid_cell_touches(const GridHeader &h, float cx, float cy, float cz, float rad, uint32_t x, uint32_t y, uint32_t z) {
    const float c[3] = {cx, cy, cz};
    const uint32_t idx[3] = {x, y, z};
    float d2 = 0.0f;
    for (int a = 0; a < 3; ++a) {
        const float lo = h.gmin[a] + (float)idx[a] * h.cell[a], hi = h.gmin[a] + (float)(idx[a] + 1u) * h.cell[a];
        const float d = c[a] < lo ? lo - c[a] : (c[a] > hi ? c[a] - hi : 0.0f);
        d2 = d2 + d * d;
    }
    const float R = rad + h.margin;
    return d2 <= R * R * 1.001f;
}
// classification key of a radius: non-finite radii sort last, so that the median (and with it the small / large split) is
// defined for every scene and the same on host and device
APT_HD float grid_radius(float r2) {
    const float rad = sqrtf(r2 > 0.0f ? r2 : 0.0f);
    return (rad == rad && rad <= 3.0e38f) ? rad : __builtin_inff();
}
APT_HD bool grid_is_large(float r2, float cx, float cy, float cz, float rad, float median) {
    const bool finite = rad <= 3.0e38f && fabsf(cx) <= 3.0e38f && fabsf(cy) <= 3.0e38f && fabsf(cz) <= 3.0e38f; // false for NaN / inf
    return !finite || rad > 8.0f * median || !(r2 >= 0.0f);
}

// ---- Russian roulette (extension, APT_FLAG_RR; specified in include/render_mi355x.h) ------------
APT_HD uint64_t rr_path_key(uint64_t seed, uint64_t path) { return splitmix64(seed ^ splitmix64(path)); }
#if defined(__clang__)
APT_HD void russian_roulette(PathState &s, uint64_t key, uint32_t bounce) { // bounce: 0-based index just shaded
    if (!s.alive) return;                       // frozen after the light: throughput no longer changes
    float q = s.rxy.x;
    if (s.rxy.y > q) q = s.rxy.y;
    if (s.rz > q) q = s.rz;
    if (!(q > 0.0f)) return;                    // already (0,0,0), negative or NaN: leave it
#if defined(__HIP_DEVICE_COMPILE__)
    const float p = __builtin_amdgcn_fmed3f(q, 0.05f, 0.95f);   // q > 0 and not NaN here: the median is clamp(q, 0.05, 0.95), one instruction
#else
    float p = q < 0.05f ? 0.05f : q;
    p = p > 0.95f ? 0.95f : p;
#endif
    const uint64_t h = splitmix64(key + 0x9E3779B97F4A7C15ull * (uint64_t)(bounce + 1u));
    const float u = (float)((uint32_t)(h >> 32) >> 8) * 0x1p-24f;   // the 24 high bits (written on the high dword: a 32-bit conversion, not a 64-bit one)
    if (u >= p) { s.rxy.x = 0.0f; s.rxy.y = 0.0f; s.rz = 0.0f; }
    else {
#if defined(__HIP_DEVICE_COMPILE__)
        // 1 / p, correctly rounded, for p in [0.05, 0.95]: hipcc's own expansion of the IEEE divide (refined v_rcp_f32, quotient, two
        // residual corrections) without its v_div_scale / v_div_fmas / v_div_fixup frame, which is the identity when neither operand
        // nor quotient comes near the ends of the exponent range (the sequence of div3_shared() with numerator 1, checked bit for bit
        // against '/' by apt_selftest_div3; the frames with roulette equal the CPU restatement's 1.0f / p bit for bit)
        const float r0 = __builtin_amdgcn_rcpf(p);
        const float e0 = __builtin_fmaf(-p, r0, 1.0f);
        const float r1 = __builtin_fmaf(e0, r0, r0);
        float e = __builtin_fmaf(-p, r1, 1.0f);
        float q = __builtin_fmaf(e, r1, r1);
        e = __builtin_fmaf(-p, q, 1.0f);
        const float inv = __builtin_fmaf(e, r1, q);
#else
        const float inv = 1.0f / p;
#endif
        s.rxy.x = s.rxy.x * inv; s.rxy.y = s.rxy.y * inv; s.rz = s.rz * inv;
    }
}

#endif // __clang__

// path index -> (i, j, sy, sx, k):  p = (((i*H + j)*2 + sy)*2 + sx)*S + k   gen_data.py:32-36
APT_HD void path_coords(uint64_t p, uint32_t H, uint32_t S, uint32_t &i, uint32_t &j, uint32_t &sy, uint32_t &sx) {
    uint64_t r = p / S;
    sx = (uint32_t)(r & 1);
    sy = (uint32_t)((r >> 1) & 1);
    r >>= 2;
    j = (uint32_t)(r % H);
    i = (uint32_t)(r / H);
}

} // namespace apt
