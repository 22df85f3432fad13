// decode_rates.hip -- R14 (decode_color, scripts/data_visualization.py:20-59) is the one HBM-bound kernel of the path: 12 bytes read per
// path, nothing else.  What read rate do different lane mappings of numpy's pairwise sum reach on the C2 colour buffer ([3][N] float32,
// N = 1920 * 1080 * 4 * 64 = 6.37 GB)?  The product kernel (decode_color_kernel8, pt_kernels.h: 8 lanes per sub-pixel row, one dword per
// lane and load) at several grid caps against decode_color_kernel4 (2 lanes per sub-pixel row with float4 loads, each owning four of numpy's
// eight accumulators; the form the product uses for 8 <= S <= 32): both must give the same bits.  Results: decode_rates_mi355x.txt.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I../../ascendpathtracing_amd/csrc -I../../include decode_rates.hip -o decode_rates && ./decode_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "render_mi355x.h"
#include "pt_kernels.h"

namespace {

__global__ void fill_kernel(float *p, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        p[i] = (float)(x & 0xffffff) * (1.0f / 16777216.0f) * 1.3f;   // [0, 1.3): some pixels clip
    }
}

}  // namespace

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char **argv) {
    const uint32_t W = 1920, H = 1080;
    const uint32_t S = argc > 1 ? (uint32_t)atoi(argv[1]) : 64;
    const uint64_t npix = (uint64_t)W * H, N = npix * 4 * S;
    LeafProg lp;
    if (!apt::make_leaf_plan(S, lp)) return 1;
    float *colors, *fb0, *fb1; uint8_t *u0, *u1;
    CK(hipMalloc(&colors, 3 * N * 4)); CK(hipMalloc(&fb0, 3 * npix * 4)); CK(hipMalloc(&fb1, 3 * npix * 4)); CK(hipMalloc(&u0, 3 * npix)); CK(hipMalloc(&u1, 3 * npix));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, colors, 3 * N);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double gb = 3.0 * N * 4 / 1e9;
    auto time = [&](const char *name, auto launch, float *fb, uint8_t *u8) {
        launch(fb, u8); launch(fb, u8); CK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int r = 0; r < 7; ++r) {
            CK(hipEventRecord(e0)); launch(fb, u8); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        printf("%-56s S=%u  %.3f ms  %.2f TB/s (%.2f of 8 TB/s peak, %.2f of the 6.3 a float4 copy reaches)\n", name, S, best, gb / best, gb / best / 8.0, gb / best / 6.3);
    };
    const uint64_t blocks8 = (npix * 3 * 4 * 8 + kBlock - 1) / kBlock;
    for (unsigned cap : {256u * 16u, 256u * 64u, 256u * 256u, 256u * 1024u, 256u * 4096u}) {   // 4096: more than the 777 600 blocks of C2 = one round
        char nm[96];
        snprintf(nm, sizeof nm, "product: 8 lanes x dword (grid cap 256*%u)", cap / 256u);
        time(nm, [&](float *fb, uint8_t *u8) { hipLaunchKernelGGL(decode_color_kernel8, dim3((unsigned)std::min<uint64_t>(blocks8, cap)), dim3(kBlock), 0, 0, colors, S, npix, lp, fb, u8); }, fb0, u0);
    }
    if (S % 4 == 0 && S <= 8u * kDecode4Blocks) {   // the sample counts the product gives decode_color_kernel4
        const uint64_t blocks4 = (npix * 3 * 4 * 2 + kBlock - 1) / kBlock;
        for (unsigned cap : {256u * 16u, 256u * 64u, 256u * 256u, 256u * 1024u}) {
            char nm[96];
            snprintf(nm, sizeof nm, "decode_color_kernel4: 2 lanes x float4 (grid cap 256*%u)", cap / 256u);
            time(nm, [&](float *fb, uint8_t *u8) { hipLaunchKernelGGL(decode_color_kernel4, dim3((unsigned)std::min<uint64_t>(blocks4, cap)), dim3(kBlock), 0, 0, colors, S, npix, lp, fb, u8); }, fb1, u1);
        }
        std::vector<uint32_t> a(3 * npix), b(3 * npix);
        std::vector<uint8_t> ua(3 * npix), ub(3 * npix);
        CK(hipMemcpy(a.data(), fb0, 3 * npix * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), fb1, 3 * npix * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ua.data(), u0, 3 * npix, hipMemcpyDeviceToHost)); CK(hipMemcpy(ub.data(), u1, 3 * npix, hipMemcpyDeviceToHost));
        size_t diff = 0, clipped = 0;
        for (size_t i = 0; i < a.size(); ++i) { diff += a[i] != b[i] || ua[i] != ub[i]; clipped += a[i] == 0x3f800000u; }
        printf("kernel4 against kernel8: %zu differing values of %zu (%zu clipped to 1)\n", diff, a.size(), clipped);
        if (diff) return 2;
    }
    return 0;
}
