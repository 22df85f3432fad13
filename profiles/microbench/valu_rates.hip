// valu_rates.hip -- issue cost (cycles per wave64 instruction per SIMD) of the VALU
// instructions the path-tracing loop is made of, on gfx950.  8 waves per SIMD, independent
// instructions, so the figure is the steady-state issue rate, not latency.
//   hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

#define KERNEL(name, body)                                                                    \
    __global__ __launch_bounds__(256) void name(float *out, int iters, float a, float b) {    \
        float v0 = a + threadIdx.x, v1 = b, v2 = a * b, v3 = a - b, v4 = v0 + 1, v5 = v1 + 2, v6 = v2 + 3, v7 = v3 + 4; \
        double d0 = a, d1 = b, d2 = a + b, d3 = a - b;                                        \
        unsigned u0 = threadIdx.x, u1 = 3;                                                    \
        for (int i = 0; i < iters; ++i) { asm volatile(REP64(body) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(u0), "+v"(u1) : "s"(a), "s"(b) : "vcc", "s20", "s21"); } \
        out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + (float)(d0 + d1 + d2 + d3) + u0 + u1; \
    }

// operands: %0-%7 f32 vgprs, %8-%11 f64 pairs, %12,%13 u32, %14,%15 sgprs
KERNEL(k_add, "v_add_f32 %0, %1, %2\n")
KERNEL(k_add_sgpr, "v_add_f32 %0, %14, %2\n")
KERNEL(k_mul, "v_mul_f32 %0, %1, %2\n")
KERNEL(k_fma, "v_fma_f32 %0, %1, %2, %3\n")
KERNEL(k_pk_mul, "v_pk_mul_f32 %8, %9, %10\n")
KERNEL(k_pk_add, "v_pk_add_f32 %8, %9, %10\n")
KERNEL(k_pk_fma, "v_pk_fma_f32 %8, %9, %10, %11\n")
KERNEL(k_mov, "v_mov_b32 %0, %1\n")
KERNEL(k_cndmask_vcc, "v_cndmask_b32 %0, %1, %2, vcc\n")
KERNEL(k_cndmask_sgpr, "v_cndmask_b32_e64 %0, %1, %2, s[20:21]\n")
KERNEL(k_cmp_vcc, "v_cmp_lt_f32 vcc, %1, %2\n")
KERNEL(k_cmp_sgpr, "v_cmp_lt_f32_e64 s[20:21], %1, %2\n")
KERNEL(k_cmp_then_cnd, "v_cmp_lt_f32 vcc, %1, %2\nv_cndmask_b32 %0, %3, %4, vcc\n")
KERNEL(k_sqrt, "v_sqrt_f32 %0, %1\n")
KERNEL(k_rcp, "v_rcp_f32 %0, %1\n")
KERNEL(k_rsq, "v_rsq_f32 %0, %1\n")
KERNEL(k_add_u32, "v_add_u32 %12, %13, %12\n")
KERNEL(k_min, "v_min_f32 %0, %1, %2\n")
KERNEL(k_min3, "v_min3_f32 %0, %1, %2, %3\n")
KERNEL(k_med3, "v_med3_f32 %0, %1, %2, %3\n")
KERNEL(k_add_f64, "v_add_f64 %8, %9, %10\n")
KERNEL(k_mul_f64, "v_mul_f64 %8, %9, %10\n")
KERNEL(k_fma_f64, "v_fma_f64 %8, %9, %10, %11\n")
KERNEL(k_cvt_f64_f32, "v_cvt_f64_f32 %8, %1\n")
KERNEL(k_cvt_f32_f64, "v_cvt_f32_f64 %0, %9\n")
KERNEL(k_nop, "s_nop 0\n")
KERNEL(k_div_scale, "v_div_scale_f32 %0, vcc, %1, %2, %1\n")
KERNEL(k_div_fmas, "v_div_fmas_f32 %0, %1, %2, %3\n")
KERNEL(k_div_fixup, "v_div_fixup_f32 %0, %1, %2, %3\n")
KERNEL(k_mul_lit, "v_mul_f32 %0, 0x40490fdb, %2\n")
KERNEL(k_sub_abs, "v_sub_f32_e64 %0, |%1|, %2\n")
KERNEL(k_mix_add_mul, "v_add_f32 %0, %1, %2\nv_mul_f32 %3, %4, %5\n")
KERNEL(k_pk_add_sgpr, "v_pk_add_f32 %8, s[20:21], %10\n")
KERNEL(k_pk_add_sgpr_neg, "v_pk_add_f32 %8, s[20:21], %10 neg_lo:[0,1] neg_hi:[0,1]\n")
KERNEL(k_pk_mul_opsel, "v_pk_mul_f32 %8, %9, %10 op_sel_hi:[1,0]\n")
KERNEL(k_min_u32, "v_min_u32 %12, %13, %12\n")
KERNEL(k_min3_u32, "v_min3_u32 %12, %13, %12, %12\n")
KERNEL(k_subrev_u32, "v_subrev_u32 %12, %13, %12\n")
KERNEL(k_cmp_u32, "v_cmp_le_u32 vcc, %12, %13\n")
KERNEL(k_pk_mov, "v_pk_mov_b32 %8, %9, %10\n")
KERNEL(k_fmac, "v_fmac_f32 %0, %1, %2\n")
KERNEL(k_mul_f32_sgpr, "v_mul_f32 %0, %14, %2\n")
KERNEL(k_add_4indep, "v_add_f32 %0, %1, %2\nv_add_f32 %3, %4, %5\nv_add_f32 %6, %7, %1\nv_add_f32 %4, %2, %5\n")
KERNEL(k_pk_4indep, "v_pk_add_f32 %8, %9, %10\nv_pk_mul_f32 %11, %9, %10\n")
KERNEL(k_mul_lo_u32, "v_mul_lo_u32 %12, %13, %12\n")
KERNEL(k_mul_hi_u32, "v_mul_hi_u32 %12, %13, %12\n")
KERNEL(k_mad_u64_u32, "v_mad_u64_u32 %8, vcc, %12, %13, %9\n")
KERNEL(k_lshr_b64, "v_lshrrev_b64 %8, 7, %9\n")
KERNEL(k_xor, "v_xor_b32 %12, %13, %12\n")
KERNEL(k_rsq_f64, "v_rsq_f64 %8, %9\n")
KERNEL(k_rcp_f64, "v_rcp_f64 %8, %9\n")
KERNEL(k_sqrt_f64, "v_sqrt_f64 %8, %9\n")
KERNEL(k_div_scale_f64, "v_div_scale_f64 %8, vcc, %9, %10, %9\n")
KERNEL(k_div_fmas_f64, "v_div_fmas_f64 %8, %9, %10, %11\n")
KERNEL(k_div_fixup_f64, "v_div_fixup_f64 %8, %9, %10, %11\n")
KERNEL(k_ldexp_f64, "v_ldexp_f64 %8, %9, %12\n")
KERNEL(k_cvt_f64_u32, "v_cvt_f64_u32 %8, %12\n")
KERNEL(k_cmp_f64, "v_cmp_lt_f64 vcc, %8, %9\n")
KERNEL(k_mul_u32_u24, "v_mul_u32_u24 %12, %13, %12\n")
KERNEL(k_rsq_3add, "v_rsq_f32 %0, %1\nv_add_f32 %2, %3, %4\nv_add_f32 %5, %6, %7\nv_add_f32 %3, %4, %6\n")
KERNEL(k_rsq_6add, "v_rsq_f32 %0, %1\nv_add_f32 %2, %3, %4\nv_add_f32 %5, %6, %7\nv_add_f32 %3, %4, %6\nv_add_f32 %2, %3, %4\nv_add_f32 %5, %6, %7\nv_add_f32 %3, %4, %6\n")
KERNEL(k_rsq_2pk, "v_rsq_f32 %0, %1\nv_pk_add_f32 %8, %9, %10\nv_pk_mul_f32 %11, %9, %10\n")
KERNEL(k_cmp_add, "v_cmp_lt_f32 vcc, %1, %2\nv_add_f32 %3, %4, %5\n")
KERNEL(k_pk_1add, "v_pk_add_f32 %8, %9, %10\nv_add_f32 %0, %1, %2\n")
KERNEL(k_pk_2add, "v_pk_mul_f32 %8, %9, %10\nv_add_f32 %0, %1, %2\nv_mul_f32 %3, %4, %5\n")
KERNEL(k_cnd_add, "v_cndmask_b32 %0, %1, %2, vcc\nv_add_f32 %3, %4, %5\n")
KERNEL(k_min_add, "v_min3_u32 %12, %13, %12, %13\nv_add_f32 %3, %4, %5\n")
KERNEL(k_f64_add, "v_fma_f64 %8, %9, %10, %11\nv_add_f32 %3, %4, %5\n")
KERNEL(k_sgpr_add, "v_add_f32 %0, %14, %2\nv_add_f32 %3, %4, %5\n")
KERNEL(k_pk_pk_add_add, "v_pk_add_f32 %8, %9, %10\nv_pk_mul_f32 %11, %9, %10\nv_add_f32 %0, %1, %2\nv_mul_f32 %3, %4, %5\n")
KERNEL(k_rsq_indep, "v_rsq_f32 %0, %1\nv_add_f32 %2, %3, %4\nv_add_f32 %5, %3, %4\nv_add_f32 %6, %3, %4\n")
KERNEL(k_rsq_cmp, "v_rsq_f32 %0, %1\nv_cmp_lt_f32 vcc, %3, %4\n")
#define F8(x) x x x x x x x x
KERNEL(k_cnd8_add8, F8("v_cndmask_b32 %0, %1, %2, vcc\n") F8("v_add_f32 %3, %4, %5\n"))
KERNEL(k_min8_add8, F8("v_min3_u32 %12, %13, %12, %13\n") F8("v_add_f32 %3, %4, %5\n"))
KERNEL(k_cnd2_add2, "v_cndmask_b32 %0, %1, %2, vcc\nv_cndmask_b32 %6, %1, %2, vcc\nv_add_f32 %3, %4, %5\nv_add_f32 %7, %4, %5\n")
KERNEL(k_pk_cnd_add, "v_pk_add_f32 %8, %9, %10\nv_cndmask_b32 %0, %1, %2, vcc\nv_add_f32 %3, %4, %5\n")
KERNEL(k_cnd_mul_dep, "v_cndmask_b32 %0, %1, %2, vcc\nv_mul_f32 %3, %0, %5\n")
KERNEL(k_readlane_like_dpp, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")

typedef void (*kfn)(float *, int, float, float);

int main() {
    float *out;
    hipMalloc(&out, sizeof(float) * 256 * 2048);
    struct T { const char *n; kfn f; int per; };
    std::vector<T> ts = {{"v_add_f32", k_add, 1}, {"v_add_f32 (sgpr src)", k_add_sgpr, 1}, {"v_mul_f32", k_mul, 1}, {"v_mul_f32 (literal)", k_mul_lit, 1},
        {"v_fma_f32", k_fma, 1}, {"v_pk_mul_f32", k_pk_mul, 1}, {"v_pk_add_f32", k_pk_add, 1}, {"v_pk_fma_f32", k_pk_fma, 1}, {"v_mov_b32", k_mov, 1},
        {"v_mov_b32 dpp", k_readlane_like_dpp, 1},
        {"v_cndmask (vcc)", k_cndmask_vcc, 1}, {"v_cndmask_e64 (sgpr)", k_cndmask_sgpr, 1}, {"v_cmp_lt (vcc)", k_cmp_vcc, 1}, {"v_cmp_lt_e64 (sgpr)", k_cmp_sgpr, 1},
        {"v_cmp+v_cndmask pair", k_cmp_then_cnd, 2}, {"v_sqrt_f32", k_sqrt, 1}, {"v_rcp_f32", k_rcp, 1}, {"v_rsq_f32", k_rsq, 1}, {"v_add_u32", k_add_u32, 1},
        {"v_min_f32", k_min, 1}, {"v_min3_f32", k_min3, 1}, {"v_med3_f32", k_med3, 1}, {"v_sub_f32 |abs| e64", k_sub_abs, 1},
        {"v_add_f64", k_add_f64, 1}, {"v_mul_f64", k_mul_f64, 1}, {"v_fma_f64", k_fma_f64, 1}, {"v_cvt_f64_f32", k_cvt_f64_f32, 1}, {"v_cvt_f32_f64", k_cvt_f32_f64, 1},
        {"v_div_scale_f32", k_div_scale, 1}, {"v_div_fmas_f32", k_div_fmas, 1}, {"v_div_fixup_f32", k_div_fixup, 1},
        {"v_add+v_mul alternating", k_mix_add_mul, 2}, {"v_pk_add + v_add (group of 2)", k_pk_1add, 2}, {"v_pk_mul + v_add + v_mul (group of 3)", k_pk_2add, 3},
        {"v_cndmask + v_add (group of 2)", k_cnd_add, 2}, {"v_min3_u32 + v_add (group of 2)", k_min_add, 2}, {"v_fma_f64 + v_add_f32 (group of 2)", k_f64_add, 2},
        {"v_add(sgpr) + v_add (group of 2)", k_sgpr_add, 2}, {"2 pk + 2 scalar (group of 4)", k_pk_pk_add_add, 4}, {"v_rsq + 3 indep v_add (group of 4)", k_rsq_indep, 4},
        {"v_rsq + v_cmp (group of 2)", k_rsq_cmp, 2}, {"v_rsq + 3 v_add (per group of 4)", k_rsq_3add, 4}, {"v_rsq + 6 v_add (per group of 7)", k_rsq_6add, 7},
        {"v_rsq + 2 pk (per group of 3)", k_rsq_2pk, 3}, {"v_cmp + v_add (per group of 2)", k_cmp_add, 2}, {"v_mul_lo_u32", k_mul_lo_u32, 1}, {"v_mul_hi_u32", k_mul_hi_u32, 1}, {"v_mad_u64_u32", k_mad_u64_u32, 1},
        {"v_mul_u32_u24", k_mul_u32_u24, 1}, {"v_lshrrev_b64", k_lshr_b64, 1}, {"v_xor_b32", k_xor, 1}, {"v_rsq_f64", k_rsq_f64, 1}, {"v_rcp_f64", k_rcp_f64, 1}, {"v_sqrt_f64", k_sqrt_f64, 1},
        {"v_div_scale_f64", k_div_scale_f64, 1}, {"v_div_fmas_f64", k_div_fmas_f64, 1}, {"v_div_fixup_f64", k_div_fixup_f64, 1}, {"v_ldexp_f64", k_ldexp_f64, 1},
        {"v_cvt_f64_u32", k_cvt_f64_u32, 1}, {"v_cmp_lt_f64", k_cmp_f64, 1}, {"v_pk_add_f32 (sgpr pair)", k_pk_add_sgpr, 1}, {"v_pk_add_f32 (sgpr pair, neg)", k_pk_add_sgpr_neg, 1},
        {"v_pk_mul_f32 op_sel_hi", k_pk_mul_opsel, 1}, {"v_min_u32", k_min_u32, 1}, {"v_min3_u32", k_min3_u32, 1}, {"v_subrev_u32", k_subrev_u32, 1},
        {"v_cmp_le_u32 (vcc)", k_cmp_u32, 1}, {"v_pk_mov_b32", k_pk_mov, 1}, {"v_fmac_f32", k_fmac, 1}, {"v_mul_f32 (sgpr src)", k_mul_f32_sgpr, 1},
        {"8 v_cndmask then 8 v_add (group of 16)", k_cnd8_add8, 16}, {"8 v_min3_u32 then 8 v_add (group of 16)", k_min8_add8, 16}, {"2 v_cndmask then 2 v_add (group of 4)", k_cnd2_add2, 4}, {"v_pk_add + v_cndmask + v_add (group of 3)", k_pk_cnd_add, 3}, {"v_cndmask + dependent v_mul (group of 2)", k_cnd_mul_dep, 2},
        {"4 independent v_add_f32", k_add_4indep, 4}, {"v_pk_add + v_pk_mul (different dst)", k_pk_4indep, 2}, {"s_nop 0", k_nop, 1}};
    const int iters = 2000, blocks = 256 * 8; // 8 blocks of 4 waves per CU -> 8 waves per SIMD
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    // clock estimate: v_add_f32 is taken to issue at 2 cycles/wave/SIMD at 8 waves/SIMD
    double add_ns = 0;
    for (auto &t : ts) {
        hipLaunchKernelGGL(t.f, dim3(blocks), dim3(256), 0, 0, out, 10, 1.5f, 0.75f);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(t.f, dim3(blocks), dim3(256), 0, 0, out, iters, 1.5f, 0.75f);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double inst_per_simd = (double)iters * 64 * t.per * 8; // 8 waves per SIMD
        double ns_per_inst = ms * 1e6 / inst_per_simd;
        if (add_ns == 0) add_ns = ns_per_inst;
        printf("%-28s %8.3f ms  %6.3f ns/inst/SIMD  = %5.2f x v_add_f32  (~%4.1f cycles @2.4GHz)\n", t.n, ms, ns_per_inst, ns_per_inst / add_ns, ns_per_inst * 2.4);
    }
    return 0;
}
