// decode_rates.hip -- R14 (decode_color, scripts/data_visualization.py:20-59) is the one HBM-bound kernel of the path: 12 bytes read per
// path, nothing else.  What read rate do different lane mappings of numpy's pairwise sum reach on the C2 colour buffer ([3][N] float32,
// N = 1920 * 1080 * 4 * 64 = 6.37 GB)?  The product kernel (decode_color_kernel8, pt_kernels.h: 8 lanes per sub-pixel row, one dword per
// lane and load) against forms with float4 loads: (v4) 2 lanes per sub-pixel row, each owning four of numpy's eight accumulators;
// every form must give the product kernel's bits.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I../../ascendpathtracing_amd/csrc -I../../include decode_rates.hip -o decode_rates && ./decode_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "render_mi355x.h"
#include "pt_kernels.h"

namespace {

constexpr int kGroups4 = kBlock / 2;   // sub-pixel groups (2 lanes) per block of the float4 form

// 2 lanes per sub-pixel row: lane L owns numpy's accumulators r[4L .. 4L+3] and reads a[8m + 4L .. 8m + 4L + 3] as ONE float4 per block m
// of 8 samples; 8 lanes = one (pixel, channel).  Same additions in the same order as pairwise_leaf / decode_color_kernel8.
template <int BATCH>
__global__ __launch_bounds__(kBlock) void decode_color_kernel_v4(const float *__restrict__ colors, uint32_t samples, uint64_t npix, LeafProg lp,
                                                                 float *__restrict__ fb, uint8_t *__restrict__ fb_u8) {
    __shared__ float stack_lds[kMaxStack * kGroups4];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t L = threadIdx.x & 1u;
    const uint32_t sub = (threadIdx.x >> 1) & 3u;
    const uint32_t slot = threadIdx.x >> 1;
    const uint64_t n_total = npix * 4 * samples;
    const uint64_t items = 3 * npix, per_block = kBlock / 8;
    const uint64_t rounds = (items + per_block * gridDim.x - 1) / (per_block * gridDim.x);
    for (uint64_t r = 0; r < rounds; ++r) {
        const uint64_t pc = (r * gridDim.x + blockIdx.x) * per_block + (threadIdx.x >> 3);
        const bool valid = pc < items;
        const uint64_t ch = valid ? pc / npix : 0, q = valid ? pc % npix : 0;
        const float *a = colors + ch * n_total + (q * 4 + sub) * samples;
        float res = 0.0f;
        uint32_t start = 0;
        int sp = 0;
        for (uint32_t leaf = 0; leaf < lp.nleaves; ++leaf) {
            const uint32_t n = lp.len(leaf), nfull = n & ~7u;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int mb = 0; mb < 16; mb += BATCH) {                     // a leaf has at most 128 samples = 16 blocks of 8
                if (8u * mb < nfull) {                                   // wave-uniform
                    float4 v[BATCH];
#pragma unroll
                    for (int m = 0; m < BATCH; ++m)
                        v[m] = (8u * (mb + m) < nfull) ? *reinterpret_cast<const float4 *>(a + start + 8u * (mb + m) + 4u * L) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int m = 0; m < BATCH; ++m) {
                        if (mb + m == 0) acc = v[0];
                        else if (8u * (mb + m) < nfull) { acc.x = acc.x + v[m].x; acc.y = acc.y + v[m].y; acc.z = acc.z + v[m].z; acc.w = acc.w + v[m].w; }
                    }
                }
            }
            float s = (acc.x + acc.y) + (acc.z + acc.w);               // (r0+r1)+(r2+r3) resp. (r4+r5)+(r6+r7)
            s = s + __shfl_xor(s, 1, 64);                               // their sum (commutative: both lanes hold the same bits)
            const uint32_t nt = n - nfull;
            if (nt) {                                                   // res += a[i] for the n % 8 trailing samples, in order
                float c[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) c[k] = (4u * L + k < nt) ? a[start + nfull + 4u * L + k] : 0.0f;
#pragma unroll
                for (int t = 0; t < 7; ++t)
                    if ((uint32_t)t < nt) s = s + __shfl(c[t & 3], (int)((lane & ~1u) + (t >> 2)), 64);
            }
            start += n;
            if (lp.nleaves == 1) {
                res = s;
            } else {
                stack_lds[sp * kGroups4 + slot] = s;
                ++sp;
                for (uint32_t m = 0; m < lp.ncomb(leaf); ++m) {
                    --sp;
                    const float x = stack_lds[(sp - 1) * kGroups4 + slot], y = stack_lds[sp * kGroups4 + slot];
                    stack_lds[(sp - 1) * kGroups4 + slot] = x + y;
                }
            }
        }
        if (lp.nleaves > 1) res = stack_lds[slot];
        const float mean = res / (float)samples;
        const int gbase = (int)(lane & ~7u);
        double acc64 = 0.0;
#pragma unroll
        for (int sq = 0; sq < 4; ++sq) acc64 = acc64 + (double)__shfl(mean, gbase + sq * 2, 64);
        const double v = acc64 / 4;
        const double cl = v < 0 ? 0 : (v > 1 ? 1 : v);
        if (valid && (lane & 7u) == 0) {
            fb[ch * npix + q] = (float)cl;
            if (fb_u8) fb_u8[q * 3 + ch] = (uint8_t)(cl * 255);
        }
    }
}

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
        printf("%-44s S=%u  %.3f ms  %.2f TB/s (%.2f of 8 TB/s peak, %.2f of the 6.3 a float4 copy reaches)\n", name, S, best, gb / best, gb / best / 8.0, gb / best / 6.3);
    };
    const uint64_t blocks8 = (npix * 3 * 4 * 8 + kBlock - 1) / kBlock;
    for (unsigned cap : {256u * 64u, 256u * 16u, 256u * 256u})
        time(cap == 256u * 64u ? "product: 8 lanes x dword (grid 256*64)" : (cap == 256u * 16u ? "product, grid 256*16" : "product, grid 256*256"),
             [&](float *fb, uint8_t *u8) { hipLaunchKernelGGL(decode_color_kernel8, dim3((unsigned)std::min<uint64_t>(blocks8, cap)), dim3(kBlock), 0, 0, colors, S, npix, lp, fb, u8); }, fb0, u0);
    if (S % 4 == 0) {
        const uint64_t blocks4 = (npix * 3 * 4 * 2 + kBlock - 1) / kBlock;
        for (unsigned cap : {256u * 16u, 256u * 32u, 256u * 64u}) {
            char nm[96];
            snprintf(nm, sizeof nm, "v4: 2 lanes x float4, 8 in flight (grid %u)", cap);
            time(nm, [&](float *fb, uint8_t *u8) { hipLaunchKernelGGL(decode_color_kernel_v4<8>, dim3((unsigned)std::min<uint64_t>(blocks4, cap)), dim3(kBlock), 0, 0, colors, S, npix, lp, fb, u8); }, fb1, u1);
            snprintf(nm, sizeof nm, "v4: 2 lanes x float4, 4 in flight (grid %u)", cap);
            time(nm, [&](float *fb, uint8_t *u8) { hipLaunchKernelGGL(decode_color_kernel_v4<4>, dim3((unsigned)std::min<uint64_t>(blocks4, cap)), dim3(kBlock), 0, 0, colors, S, npix, lp, fb, u8); }, fb1, u1);
            snprintf(nm, sizeof nm, "v4: 2 lanes x float4, 16 in flight (grid %u)", cap);
            time(nm, [&](float *fb, uint8_t *u8) { hipLaunchKernelGGL(decode_color_kernel_v4<16>, dim3((unsigned)std::min<uint64_t>(blocks4, cap)), dim3(kBlock), 0, 0, colors, S, npix, lp, fb, u8); }, fb1, u1);
        }
        std::vector<uint32_t> a(3 * npix), b(3 * npix);
        std::vector<uint8_t> ua(3 * npix), ub(3 * npix);
        CK(hipMemcpy(a.data(), fb0, 3 * npix * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), fb1, 3 * npix * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ua.data(), u0, 3 * npix, hipMemcpyDeviceToHost)); CK(hipMemcpy(ub.data(), u1, 3 * npix, hipMemcpyDeviceToHost));
        size_t diff = 0, clipped = 0;
        for (size_t i = 0; i < a.size(); ++i) { diff += a[i] != b[i] || ua[i] != ub[i]; clipped += a[i] == 0x3f800000u; }
        printf("v4 against the product kernel: %zu differing values of %zu (%zu clipped to 1)\n", diff, a.size(), clipped);
        if (diff) return 2;
    }
    return 0;
}
