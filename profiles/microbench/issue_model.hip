// issue_model.hip -- how the gfx950 VALU issue rate depends on occupancy and on dependences inside a wave.
// For W = 1..8 waves per SIMD: cycles per wave64 instruction per SIMD of
//   plain4   4 independent v_add_f32          plaindep  one dependent chain of v_add_f32
//   pk2      2 independent v_pk_add_f32       pkdep     one dependent chain of v_pk_add_f32 (with the s_nop the hazard needs)
//   half4    4 independent v_min_f32          mixed     the bounce block's rough mix: 4 pk, 2 plain, 1 min3, 1 cmp, (1 rsq per 3 groups)
//   hipcc -O3 --offload-arch=gfx950 issue_model.hip -o issue_model && ./issue_model
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)
#define KERNEL(name, body)                                                                                         \
    __global__ __launch_bounds__(256) void name(float *out, int iters, float a, float b) {                         \
        float v0 = a + threadIdx.x, v1 = b, v2 = a * b, v3 = a - b, v4 = v0 + 1, v5 = v1 + 2, v6 = v2 + 3, v7 = v3 + 4; \
        double d0 = a, d1 = b, d2 = a + b, d3 = a - b;                                                             \
        unsigned u0 = threadIdx.x, u1 = 3;                                                                         \
        for (int i = 0; i < iters; ++i) { asm volatile(REP32(body) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(u0), "+v"(u1) : : "vcc"); } \
        out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + (float)(d0 + d1 + d2 + d3) + u0 + u1; \
    }
KERNEL(k_plain4, "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %6\n v_add_f32 %3, %3, %7\n")
KERNEL(k_plaindep, "v_add_f32 %0, %0, %4\n v_add_f32 %0, %0, %5\n v_add_f32 %0, %0, %6\n v_add_f32 %0, %0, %7\n")
KERNEL(k_pk2, "v_pk_add_f32 %8, %8, %10\n v_pk_add_f32 %9, %9, %11\n")
KERNEL(k_pkdep, "v_pk_add_f32 %8, %8, %10\n s_nop 0\n v_pk_add_f32 %8, %8, %11\n s_nop 0\n")
KERNEL(k_half4, "v_min_f32 %0, %0, %4\n v_min_f32 %1, %1, %5\n v_min_f32 %2, %2, %6\n v_min_f32 %3, %3, %7\n")
KERNEL(k_mixed, "v_pk_mul_f32 %8, %10, %11\n v_pk_add_f32 %9, %10, %11\n v_sub_u32 %12, %12, %13\n v_pk_mul_f32 %10, %8, %9\n v_min3_u32 %13, %13, %12, %12\n v_pk_add_f32 %11, %8, %9\n v_add_f32 %0, %1, %2\n v_cmp_ne_u32 vcc, %12, %13\n")
KERNEL(k_trans, "v_rsq_f32 %0, %4\n v_rsq_f32 %1, %5\n")
KERNEL(k_f64, "v_fma_f64 %8, %10, %11, %8\n v_fma_f64 %9, %10, %11, %9\n")
KERNEL(k_mullo, "v_mul_lo_u32 %12, %12, %13\n v_mul_lo_u32 %13, %13, %12\n")
KERNEL(k_mad64, "v_mad_u64_u32 %8, vcc, %12, %13, %8\n v_mad_u64_u32 %9, vcc, %13, %12, %9\n")
KERNEL(k_mul24, "v_mul_u32_u24 %12, %12, %13\n v_mul_u32_u24 %13, %13, %12\n")
KERNEL(k_add64, "v_lshl_add_u64 %8, %8, 0, %10\n v_lshl_add_u64 %9, %9, 0, %11\n")
KERNEL(k_shr64, "v_lshrrev_b64 %8, 30, %8\n v_lshrrev_b64 %9, 27, %9\n")
KERNEL(k_mulf64, "v_mul_f64 %8, %10, %11\n v_add_f64 %9, %10, %11\n")
KERNEL(k_rcp64, "v_rcp_f64 %8, %10\n v_rsq_f64 %9, %11\n")
KERNEL(k_cvt, "v_cvt_f32_f64 %0, %8\n v_cvt_f64_f32 %9, %1\n")
typedef void (*kfn)(float *, int, float, float);
int main() {
    float *out;
    hipMalloc(&out, sizeof(float) * 256 * 2048);
    struct T { const char *n; kfn f; int per; } ts[] = {{"plain4", k_plain4, 4}, {"plaindep", k_plaindep, 4}, {"pk2", k_pk2, 2}, {"pkdep", k_pkdep, 2},
                                                       {"half4", k_half4, 4}, {"mixed(8)", k_mixed, 8}, {"trans2", k_trans, 2}, {"f64fma2", k_f64, 2},
                                                       {"mul_lo_u32", k_mullo, 2}, {"mad_u64_u32", k_mad64, 2}, {"mul_u32_u24", k_mul24, 2}, {"lshl_add_u64", k_add64, 2},
                                                       {"lshrrev_b64", k_shr64, 2}, {"mul/add_f64", k_mulf64, 2}, {"rcp/rsq_f64", k_rcp64, 2}, {"cvt f32<>f64", k_cvt, 2}};
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    printf("cycles per wave64 VALU instruction per SIMD at 2.4 GHz (lower = faster); W = waves per SIMD\n%-13s", "kernel");
    for (int w = 1; w <= 8; ++w) printf("   W=%d ", w);
    printf("\n");
    const int iters = 4000;
    for (auto &t : ts) {
        printf("%-13s", t.n);
        for (int w = 1; w <= 8; ++w) {
            const int blocks = 256 * w; // one 4-wave block per CU per W -> W waves on every SIMD
            hipLaunchKernelGGL(t.f, dim3(blocks), dim3(256), 0, 0, out, 10, 1.5f, 0.75f);
            hipDeviceSynchronize();
            hipEventRecord(a);
            hipLaunchKernelGGL(t.f, dim3(blocks), dim3(256), 0, 0, out, iters, 1.5f, 0.75f);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double inst_per_simd = (double)iters * 32 * t.per * w;
            printf(" %6.2f ", ms * 1e6 / inst_per_simd * 2.4);
        }
        printf("\n");
    }
    return 0;
}
