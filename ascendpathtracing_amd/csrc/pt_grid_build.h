// pt_grid_build.h -- apt_build_grid_device: the uniform grid of pt_core.h (GridHeader) built ON THE DEVICE from the
// [10][Ns] table in HBM, byte-identical to what apt_build_grid_host writes for the same scene (included by
// render_kernels.hip only).  The scalar header arithmetic (margin, cell counts from the cube root of the volume, ...)
// is the host's own (grid_header_from_stats, shared with apt_build_grid_host), fed with statistics the device reduces;
// everything that is O(Ns) or O(cells) runs in kernels:
//   1  radii + exact median by radix select on the float bit patterns (4 passes of 8 bits, one workgroup)
//   2  small / large classification, ordered lists by a scan, bounding box of the small spheres   -> 40-byte read-back
//   3  cell counts (atomics), exclusive scan over the cells                                          -> 4-byte read-back
//   4  items scattered through per-cell cursors, every cell's list sorted (ascending sphere index, as the host's: short
//      lists on the device, lists longer than 32 items on the host), geometry tables, pair-slot tables
// Synchronous on `stream` (two small read-backs size the buffer); a build step, not a render call.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "pt_core.h"

namespace {

using namespace apt;

constexpr int kGB = 1024; // threads of the single-workgroup kernels

struct GridBuildStats { // device -> host after phase 2
    uint32_t nsmall, nlarge;
    float lo[3], hi[3], scale;
    float median;
};

// rad[k]; the median = element ns/2 of the sorted radii (what std::nth_element yields), by radix select; then the
// classification, the two ordered lists and the small spheres' bounds.  One workgroup: Ns is 1e4..1e6 here.
__global__ __launch_bounds__(kGB) void grid_classify_kernel(const float *__restrict__ sph, uint32_t ns, float *__restrict__ rad,
                                                           uint32_t *__restrict__ large, uint32_t *__restrict__ small,
                                                           GridBuildStats *__restrict__ st) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_prefix, s_rank, s_base[2];
    __shared__ uint32_t scan_s[kGB], scan_l[kGB];
    __shared__ float red[7][kGB / 64];
    const uint32_t t = threadIdx.x;
    const float *r2 = sph, *cx = sph + ns, *cy = sph + 2 * (size_t)ns, *cz = sph + 3 * (size_t)ns;
    for (uint32_t k = t; k < ns; k += kGB) rad[k] = grid_radius(r2[k]);
    if (t == 0) { s_prefix = 0; s_rank = ns / 2; }
    __syncthreads();
    for (int pass = 3; pass >= 0; --pass) { // radii are >= 0 (or +inf): their bit patterns order like the values
        for (uint32_t i = t; i < 256; i += kGB) hist[i] = 0;
        __syncthreads();
        const uint32_t shift = 8u * pass, prefix = s_prefix;
        const uint32_t mask_hi = pass == 3 ? 0u : (0xffffffffu << (shift + 8));
        for (uint32_t k = t; k < ns; k += kGB) {
            const uint32_t b = __float_as_uint(rad[k]);
            if ((b & mask_hi) == prefix) atomicAdd(&hist[(b >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (t == 0) {
            uint32_t rank = s_rank, d = 0;
            for (; d < 256; ++d) { if (rank < hist[d]) break; rank -= hist[d]; }
            s_prefix = prefix | (d << shift);
            s_rank = rank;
        }
        __syncthreads();
    }
    const float median = __uint_as_float(s_prefix);
    // ordered compaction: chunks of kGB spheres, exclusive scans of the two flags inside the chunk, running bases
    if (t == 0) { s_base[0] = 0; s_base[1] = 0; }
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f}, scale = 1.0f;
    __syncthreads();
    for (uint32_t base = 0; base < ns; base += kGB) {
        const uint32_t k = base + t;
        const bool in = k < ns;
        const bool lg = in && grid_is_large(r2[k], cx[k], cy[k], cz[k], rad[k], median);
        const bool sm = in && !lg;
        scan_s[t] = sm; scan_l[t] = lg;
        __syncthreads();
        for (uint32_t off = 1; off < kGB; off <<= 1) { // Hillis-Steele inclusive scan of both arrays
            const uint32_t a = t >= off ? scan_s[t - off] : 0u, b = t >= off ? scan_l[t - off] : 0u;
            __syncthreads();
            scan_s[t] += a; scan_l[t] += b;
            __syncthreads();
        }
        if (sm) small[s_base[0] + scan_s[t] - 1] = k;
        if (lg) large[s_base[1] + scan_l[t] - 1] = k;
        if (sm) {
            const float c[3] = {cx[k], cy[k], cz[k]};
#pragma unroll
            for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], c[a] - rad[k]); hi[a] = fmaxf(hi[a], c[a] + rad[k]); scale = fmaxf(scale, fabsf(c[a]) + rad[k]); }
        }
        __syncthreads();
        if (t == kGB - 1) { s_base[0] += scan_s[t]; s_base[1] += scan_l[t]; }
        __syncthreads();
    }
    // min / max reductions (exact, order independent)
    float v[7] = {lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], scale};
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        for (int off = 32; off > 0; off >>= 1) {
            const float o = __shfl_xor(v[q], off, 64);
            v[q] = q < 3 ? fminf(v[q], o) : fmaxf(v[q], o);
        }
        if ((t & 63) == 0) red[q][t >> 6] = v[q];
    }
    __syncthreads();
    if (t == 0) {
        for (int q = 0; q < 7; ++q) {
            float r = red[q][0];
            for (int w = 1; w < kGB / 64; ++w) r = q < 3 ? fminf(r, red[q][w]) : fmaxf(r, red[q][w]);
            v[q] = r;
        }
        st->nsmall = s_base[0]; st->nlarge = s_base[1];
        for (int a = 0; a < 3; ++a) { st->lo[a] = v[a]; st->hi[a] = v[3 + a]; }
        st->scale = v[6];
        st->median = median;
    }
}

__global__ __launch_bounds__(256) void grid_count_kernel(GridHeader h, const float *__restrict__ sph, const float *__restrict__ rad,
                                                         const uint32_t *__restrict__ small, uint32_t nsmall,
                                                         uint32_t *__restrict__ count /* [ncells+1], zeroed */) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nsmall) return;
    const uint32_t ns = h.num_spheres, k = small[i];
    uint32_t x0, x1, y0, y1, z0, z1;
    grid_cell_range(h, sph[ns + k], rad[k], 0, x0, x1);
    grid_cell_range(h, sph[2 * (size_t)ns + k], rad[k], 1, y0, y1);
    grid_cell_range(h, sph[3 * (size_t)ns + k], rad[k], 2, z0, z1);
    const float cx = sph[ns + k], cy = sph[2 * (size_t)ns + k], cz = sph[3 * (size_t)ns + k];
    for (uint32_t z = z0; z <= z1; ++z) for (uint32_t y = y0; y <= y1; ++y) for (uint32_t x = x0; x <= x1; ++x)
        if (grid_cell_touches(h, cx, cy, cz, rad[k], x, y, z)) atomicAdd(&count[(z * h.n[1] + y) * h.n[0] + x + 1], 1u);
}

// exclusive scan in place over n words: per-block sums, a one-block scan of the sums, add-back (three launches)
__global__ __launch_bounds__(kGB) void scan_blocks_kernel(uint32_t *__restrict__ data, uint64_t n, uint32_t *__restrict__ sums) {
    __shared__ uint32_t s[kGB];
    const uint64_t i = (uint64_t)blockIdx.x * kGB + threadIdx.x;
    const uint32_t t = threadIdx.x;
    s[t] = i < n ? data[i] : 0u;
    __syncthreads();
    for (uint32_t off = 1; off < kGB; off <<= 1) {
        const uint32_t a = t >= off ? s[t - off] : 0u;
        __syncthreads();
        s[t] += a;
        __syncthreads();
    }
    if (i < n) data[i] = s[t];                 // inclusive inside the block
    if (t == kGB - 1) sums[blockIdx.x] = s[t];
}
__global__ __launch_bounds__(kGB) void scan_sums_kernel(uint32_t *__restrict__ sums, uint32_t nblocks) { // one block, sequential chunks
    __shared__ uint32_t s[kGB];
    __shared__ uint32_t carry;
    const uint32_t t = threadIdx.x;
    if (t == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nblocks; base += kGB) {
        s[t] = base + t < nblocks ? sums[base + t] : 0u;
        __syncthreads();
        for (uint32_t off = 1; off < kGB; off <<= 1) {
            const uint32_t a = t >= off ? s[t - off] : 0u;
            __syncthreads();
            s[t] += a;
            __syncthreads();
        }
        if (base + t < nblocks) sums[base + t] = carry + s[t]; // inclusive, with the carry of the chunks before
        __syncthreads();
        if (t == kGB - 1) carry += s[t];
        __syncthreads();
    }
}
__global__ __launch_bounds__(kGB) void scan_add_kernel(uint32_t *__restrict__ data, uint64_t n, const uint32_t *__restrict__ sums) {
    const uint64_t i = (uint64_t)blockIdx.x * kGB + threadIdx.x;
    if (i < n && blockIdx.x > 0) data[i] += sums[blockIdx.x - 1];
}

__global__ __launch_bounds__(256) void grid_fill_kernel(GridHeader h, const float *__restrict__ sph, const float *__restrict__ rad,
                                                        const uint32_t *__restrict__ small, uint32_t nsmall,
                                                        const uint32_t *__restrict__ cell_start, uint32_t *__restrict__ cursor /* [ncells], zeroed */,
                                                        uint32_t *__restrict__ items) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nsmall) return;
    const uint32_t ns = h.num_spheres, k = small[i];
    uint32_t x0, x1, y0, y1, z0, z1;
    grid_cell_range(h, sph[ns + k], rad[k], 0, x0, x1);
    grid_cell_range(h, sph[2 * (size_t)ns + k], rad[k], 1, y0, y1);
    grid_cell_range(h, sph[3 * (size_t)ns + k], rad[k], 2, z0, z1);
    const float cx = sph[ns + k], cy = sph[2 * (size_t)ns + k], cz = sph[3 * (size_t)ns + k];
    for (uint32_t z = z0; z <= z1; ++z) for (uint32_t y = y0; y <= y1; ++y) for (uint32_t x = x0; x <= x1; ++x) {
        if (!grid_cell_touches(h, cx, cy, cz, rad[k], x, y, z)) continue;
        const uint32_t c = (z * h.n[1] + y) * h.n[0] + x;
        items[cell_start[c] + atomicAdd(&cursor[c], 1u)] = k;
    }
}
// Ascending sphere index inside every cell (the host fills in that order).  Lists are normally a few entries long and are
// sorted here, one thread per cell; a cell with more than kGridSortInline items (clustered or coincident spheres: 1e5 spheres
// in one spot make lists of 1e5) is only REPORTED -- (begin, end) appended to `long_cells` -- and sorted by the host
// afterwards (apt_build_grid_device: one copy of the item array out and back, std::sort per reported cell): a single-thread
// insertion sort over such a list would be 1e10 dependent global accesses, i.e. a hung GPU.  At most nitems / (kGridSortInline + 1)
// cells can be long, which sizes the list.
constexpr uint32_t kGridSortInline = 32;
__global__ __launch_bounds__(256) void grid_sort_cells_kernel(uint32_t ncells, const uint32_t *__restrict__ cell_start, uint32_t *__restrict__ items,
                                                              uint32_t *__restrict__ long_count, uint2 *__restrict__ long_cells) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= ncells) return;
    const uint32_t b = cell_start[c], e = cell_start[c + 1];
    if (e - b > kGridSortInline) { long_cells[atomicAdd(long_count, 1u)] = make_uint2(b, e); return; }
    for (uint32_t i = b + 1; i < e; ++i) {
        const uint32_t v = items[i];
        uint32_t j = i;
        for (; j > b && items[j - 1] > v; --j) items[j] = items[j - 1];
        items[j] = v;
    }
}
__global__ __launch_bounds__(256) void grid_geom_kernel(const float *__restrict__ sph, uint32_t ns, const uint32_t *__restrict__ items,
                                                        uint32_t nitems, float4 *__restrict__ geom, float4 *__restrict__ item_geom) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < ns) geom[i] = make_float4(sph[ns + i], sph[2 * (size_t)ns + i], sph[3 * (size_t)ns + i], sph[i]);
    if (i < nitems) { const uint32_t k = items[i]; item_geom[i] = make_float4(sph[ns + k], sph[2 * (size_t)ns + k], sph[3 * (size_t)ns + k], sph[k]); }
}

// pair-slot tables (pt_core.h): one thread per entry of the bordered cellslot table, one more for the always-tested list; after grid_geom_kernel
__global__ __launch_bounds__(256) void grid_slots_kernel(uint32_t *__restrict__ w, GridHeader h, const float *__restrict__ sph) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c <= grid_bordered_cells(h.n)) grid_fill_cell_slots(w, h, c);
    if (c < h.num_spheres) grid_fill_sphere8(w, h, sph, c);
}

} // namespace
