// ta_rates.hip -- what a vector-memory load costs the CU's address path (TA / vL1D) on gfx950 as a function of the number of
// ACTIVE lanes, the access width and the address pattern.  The grid walk of the large-scene path (pt_trace.h grid_segment) runs
// with ~30 % of its lanes active and its TA 90 % busy (profiles/history/r03_grid_ta_pmc.json): whether a load of a half-empty wave costs
// half decides whether filling the lanes can pay.
//   hipcc -O3 --offload-arch=gfx950 ta_rates.hip -o ta_rates && ./ta_rates
// Every wave runs 4 independent address chains (an LCG per chain, no dependence on loaded data), 8 waves per SIMD, 8 blocks of
// 256 threads per CU: throughput, not latency.  Output: CU cycles per wave-level load instruction (2.4 GHz assumed).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

template <int W> struct Vec;
template <> struct Vec<1> { using T = uint32_t; static __device__ uint32_t fold(T v) { return v; } };
template <> struct Vec<2> { using T = uint2; static __device__ uint32_t fold(T v) { return v.x ^ v.y; } };
template <> struct Vec<4> { using T = uint4; static __device__ uint32_t fold(T v) { return v.x ^ v.y ^ v.z ^ v.w; } };

// pattern 0: every lane its own random element; 1: consecutive elements (coalesced); 2: one element for the whole wave
template <int W>
__global__ __launch_bounds__(256) void loads(const uint32_t *tab, uint32_t mask, int iters, int nactive, int spread, int pattern, uint32_t *out) {
    const uint32_t lane = threadIdx.x & 63u;
    const bool active = spread ? (lane % (64u / (uint32_t)nactive) == 0u) : lane < (uint32_t)nactive;
    if (!active) return;
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t wave = gid >> 6;
    uint32_t s[4], acc = 0;
    for (int c = 0; c < 4; ++c) s[c] = (pattern == 0 ? gid : wave) * 2654435761u + (uint32_t)c * 40503u + 12345u;
    const typename Vec<W>::T *t = reinterpret_cast<const typename Vec<W>::T *>(tab);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            s[c] = s[c] * 1664525u + 1013904223u;
            uint32_t idx = s[c] >> 8;
            if (pattern == 1) idx += lane;
            acc ^= Vec<W>::fold(t[idx & mask]);
        }
    }
    out[gid] = acc;
}

int main() {
    const int cus = 256, blocks = cus * 8, iters = 1000;
    uint32_t *tab, *out;
    const size_t tab_bytes = 8u << 20;
    hipMalloc(&tab, tab_bytes);
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    std::vector<uint32_t> h(tab_bytes / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)i * 2654435761u;
    hipMemcpy(tab, h.data(), tab_bytes, hipMemcpyHostToDevice);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const char *pat[3] = {"random per lane", "consecutive", "one element"};
    printf("%-18s %-6s %-10s %-8s %10s\n", "pattern", "width", "table", "lanes", "CU cycles per wave-level load");
    for (int pattern = 0; pattern < 3; ++pattern)
        for (int w = 1; w <= 4; w *= 2)
            for (size_t tbytes : {(size_t)16 << 10, (size_t)512 << 10}) {
                const uint32_t mask = (uint32_t)(tbytes / (4 * w)) - 1u;
                for (int spread = 0; spread < 2; ++spread)
                    for (int n : {64, 32, 16, 8, 4, 1}) {
                        if (spread && (n == 64 || n == 1)) continue;
                        float best = 1e30f;
                        for (int rep = 0; rep < 3; ++rep) {
                            hipEventRecord(a);
                            if (w == 1) hipLaunchKernelGGL(loads<1>, dim3(blocks), dim3(256), 0, 0, tab, mask, iters, n, spread, pattern, out);
                            if (w == 2) hipLaunchKernelGGL(loads<2>, dim3(blocks), dim3(256), 0, 0, tab, mask, iters, n, spread, pattern, out);
                            if (w == 4) hipLaunchKernelGGL(loads<4>, dim3(blocks), dim3(256), 0, 0, tab, mask, iters, n, spread, pattern, out);
                            hipEventRecord(b);
                            hipEventSynchronize(b);
                            float ms; (void)hipEventElapsedTime(&ms, a, b);
                            if (rep && ms < best) best = ms;
                        }
                        const double wave_loads_per_cu = 8.0 * 4 * iters * 4; // 8 blocks x 4 waves x iters x 4 chains
                        printf("%-18s dwordx%d %6zu KB  %2d %-6s %10.2f\n", pat[pattern], w, tbytes >> 10, n, spread ? "spread" : "low", best * 1e-3 * 2.4e9 / wave_loads_per_cu);
                        fflush(stdout);
                    }
            }
    return 0;
}
