// issue_slots.hip -- does a scalar instruction cost a wave one of its issue slots?  (why the grid form of the
// sample-queue kernel, 0.59 scalar instructions per vector one, wants resident waves.)  For W = 1..8 waves per SIMD: cycles per wave64
// VALU instruction per SIMD of a stream of one v_pk_add_f32 followed by K independent scalar ALU instructions (K = 0, 1, 2, 4), and of
// the same with s_nop 0 in place of the scalar instructions.
//   hipcc -O3 --offload-arch=gfx950 issue_slots.hip -o issue_slots && ./issue_slots
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)
#define KERNEL(name, body)                                                                                          \
    __global__ __launch_bounds__(256) void name(float *out, int iters, float a, float b) {                          \
        typedef float f2 __attribute__((ext_vector_type(2)));                                                      \
        f2 v0 = {a + threadIdx.x, b}, v1 = {a * b, a - b}, v2 = {b, a}, v3 = {a, a};                               \
        unsigned long long s0 = 0x1234567ull + iters, s1 = 0x7654321ull, s2 = 3, s3 = 5;                           \
        for (int i = 0; i < iters; ++i) { asm volatile(REP32(body) : "+v"(v0), "+v"(v1), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(v2), "v"(v3) : "scc"); } \
        out[blockIdx.x * 256 + threadIdx.x] = v0.x + v0.y + v1.x + v1.y + (float)(s0 ^ s1 ^ s2 ^ s3);            \
    }
#define PK "v_pk_add_f32 %0, %0, %6\n v_pk_add_f32 %1, %1, %7\n"
KERNEL(k_s0, PK)
KERNEL(k_s1, "v_pk_add_f32 %0, %0, %6\n s_and_b64 %2, %2, %3\n v_pk_add_f32 %1, %1, %7\n s_or_b64 %4, %4, %5\n")
KERNEL(k_s2, "v_pk_add_f32 %0, %0, %6\n s_and_b64 %2, %2, %3\n s_or_b64 %4, %4, %5\n v_pk_add_f32 %1, %1, %7\n s_andn2_b64 %3, %3, %2\n s_xor_b64 %5, %5, %4\n")
KERNEL(k_s4, "v_pk_add_f32 %0, %0, %6\n s_and_b64 %2, %2, %3\n s_or_b64 %4, %4, %5\n s_andn2_b64 %3, %3, %2\n s_xor_b64 %5, %5, %4\n"
             "v_pk_add_f32 %1, %1, %7\n s_and_b64 %2, %2, %3\n s_or_b64 %4, %4, %5\n s_andn2_b64 %3, %3, %2\n s_xor_b64 %5, %5, %4\n")
KERNEL(k_n1, "v_pk_add_f32 %0, %0, %6\n s_nop 0\n v_pk_add_f32 %1, %1, %7\n s_nop 0\n")
KERNEL(k_n2, "v_pk_add_f32 %0, %0, %6\n s_nop 0\n s_nop 0\n v_pk_add_f32 %1, %1, %7\n s_nop 0\n s_nop 0\n")
typedef void (*kfn)(float *, int, float, float);
int main() {
    float *out;
    hipMalloc(&out, sizeof(float) * 256 * 2048);
    struct T { const char *n; kfn f; } ts[] = {{"pk alone", k_s0}, {"pk + 1 salu", k_s1}, {"pk + 2 salu", k_s2}, {"pk + 4 salu", k_s4}, {"pk + 1 s_nop", k_n1}, {"pk + 2 s_nop", k_n2}};
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    printf("cycles per wave64 v_pk_add_f32 per SIMD at 2.4 GHz (lower = faster); W = waves per SIMD\n%-14s", "stream");
    for (int w = 1; w <= 8; ++w) printf("   W=%d ", w);
    printf("\n");
    const int iters = 4000;
    for (auto &t : ts) {
        printf("%-14s", t.n);
        for (int w = 1; w <= 8; ++w) {
            const int blocks = 256 * w; // one 4-wave block per CU per W -> W waves on every SIMD
            hipLaunchKernelGGL(t.f, dim3(blocks), dim3(256), 0, 0, out, 10, 1.5f, 0.75f);
            hipDeviceSynchronize();
            hipEventRecord(a);
            hipLaunchKernelGGL(t.f, dim3(blocks), dim3(256), 0, 0, out, iters, 1.5f, 0.75f);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double inst_per_simd = (double)iters * 32 * 2 * w;
            printf(" %6.2f ", ms * 1e6 / inst_per_simd * 2.4);
        }
        printf("\n");
    }
    return 0;
}
