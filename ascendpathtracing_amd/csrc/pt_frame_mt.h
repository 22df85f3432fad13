// pt_frame_mt.h -- the reference's EXACT pipeline in ONE kernel: MT19937 gen_rays (scripts/gen_data.py:21-75,438) -> the
// render loop (src/render.cpp:104-207 / gen_data.py:246-429) -> decode_color (scripts/data_visualization.py:36-57), with no
// [6][N] ray buffer and no [3][N] colour buffer in HBM ("removes the rays.bin file/HBM boundary").  Included
// by pt_kernels.h.  Round 2 ran this pipeline as three kernels with 36 bytes per path through HBM (19 GB at C2).
//
// np.random.rand() takes two MT19937 words per double and gen_rays two doubles per path, in path order, so the generator's
// output words [4 p, 4 p + 4) belong to path p and output block b (624 words) is exactly paths [156 b, 156 b + 156).  With S a
// power of two, 78 pixels = 78 * 4 S = 312 S paths are exactly 2 S blocks (and, for S >= 64, a whole number of 512-path rounds):
// one workgroup per such pixel GROUP, started from a host-made raw generator state of block g * 2 S
// (apt_mt19937_checkpoints_window with stride 2 S).  The workgroup
//   * extends the generator's RAW word sequence y[n + 624] = f(y[n], y[n + 1], y[n + 397]) in a 4096-word ring in LDS, 227 new
//     words at a time (the largest chunk whose inputs are all older than the chunk) and 227 + 227 + 169 = 623 words per barrier: the
//     far operand of a thread's second (third) word is the first (second) word it has just made; tempering happens in the registers of the thread that
//     consumes a word,
//   * works through its 312 S paths in rounds of 512 consecutive paths, TWO per thread (t and 256 + t: the two-paths-per-lane
//     bounce of pt_trace2.h): float64 camera maths, all bounces, colour = throughput * gain,
//   * puts the 512 colours of a round into LDS, where each run of S consecutive samples (one sub-pixel) is summed exactly as
//     numpy's pairwise np.mean sums it (8 lanes per run and channel: lane j owns accumulator r[j], 3-step butterfly; two leaves
//     of 128 for S = 256), and keeps the 312 sub-pixel means of its 78 pixels,
//   * finally adds the four sub-pixel means of every pixel in float64, clips, and writes float and 8-bit pixels.
// Same arithmetic, operation for operation, as gen_rays_mt_kernel -> render_paths_kernel -> decode_color_kernel8: the frame is
// bit-identical to that pipeline's and (O-mode) to the reference's own scripts (tests/test_gpu_parity.py, test_fullsize_hashes.py).
#pragma once
#include "pt_trace.h"
#include "pt_trace2.h"
#include "pt_queue.h" // FrameArgs

namespace {

constexpr uint32_t kMtGroupPixels = 78;      // 78 pixels * 4 S paths = 2 S generator blocks of 156 paths (S a power of two)
constexpr uint32_t kMtRing = 4096;           // raw generator words kept in LDS (a round reads 2048, the recurrence reaches 624 back)
constexpr uint32_t kMtRound = 512;           // paths per round: two per thread
constexpr uint32_t kMtStep = 227;            // new words per chunk: y[n + 624] needs y[n + 397], so 624 - 397 at a time
constexpr uint32_t kMtThird = 169;           // threads that can make a third word in the same step (624 - 455)

__device__ __forceinline__ uint32_t mt_temper_word(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

struct MtFrameArgs {
    const uint32_t *checkpoints; // [groups][624]: raw state of block (first_group + k) * 2 S
    uint64_t first_group;        // pixel group of checkpoints[0]
    uint32_t log2_s;             // S = 1 << log2_s, 8 <= S <= 256
};

template <int MODE>
__global__ __launch_bounds__(kBlock, APT_TWO_WAVES) void render_frame_mt_kernel(const float *__restrict__ sph, FrameArgs fa, TraceArgs ta, MtFrameArgs ma) {
    __shared__ float4 tab[kTab8Floats4];
    __shared__ Camera cam;
    __shared__ __align__(16) uint32_t ring[kMtRing];          // y[n] at ring[n % 4096]
    __shared__ float cbuf[3 * kMtRound];                       // [channel][512]: the colours of a round, in path order
    __shared__ float submean[kMtGroupPixels * 4 * 3];          // [pixel of the group][sub-pixel][channel]
    const uint32_t t = threadIdx.x;
    if (t < sizeof(Camera) / sizeof(double)) (&cam.pos[0])[t] = (&fa.cam.pos[0])[t];
    Scene8 sc;
    const Tab8 tab8 = load_scene8(sph, sc, tab);                // ends with a barrier
    const bool planes = sc.planes;
    const Gain3 gain = load_gain(sph, ta);
    const uint32_t S = 1u << ma.log2_s, H = fa.height;
    const uint32_t blk = xcd_chunked_block<2>(blockIdx.x, gridDim.x);   // (XCD-aware: neighbouring pixel groups through one L2)
    const uint64_t group = ma.first_group + blk;
    const uint32_t npaths = 312u * S;                           // of this group
    const uint64_t q0 = group * kMtGroupPixels;                 // first pixel
    const uint32_t i0 = (uint32_t)(q0 / H), j0 = (uint32_t)(q0 % H);
    const uint64_t pix_end = fa.pixel_begin + fa.pixel_count;
    for (uint32_t i = t; i < 624; i += kBlock) ring[i] = ma.checkpoints[(uint64_t)blk * 624 + i];   // y[0 .. 623] = the state of the group's first block
    __syncthreads();

    uint32_t have = 624;                                        // words y[0 .. have) exist
    uint32_t traced = 0;
    const uint32_t nrounds = (npaths + kMtRound - 1) / kMtRound;
    for (uint32_t r = 0; r < nrounds; ++r) {
        // ---- extend the raw sequence to the words this round reads: [2048 r, 2048 r + 2048) ----
        // (ring safety: a step writes y[have .. have + 623) over y[have - 4096 ..), and the oldest word still needed is
        //  min(2048 r, have - 624): with have <= 2048 (r + 1) + 622 that is less than 2671 words back)
        const uint32_t need = min(4u * npaths, 2048u * (r + 1u));
        while (have < need) {
            if (t < kMtStep) {
                // y[n + 624] = y[n + 397] ^ twist(y[n], y[n + 1]).  Thread t makes y[nn + 624] and then y[nn + 227 + 624] as well: its "397
                // back" operand, y[nn + 227 + 397], IS the word the thread has just made, the other two are old words -- two chunks
                // of 227 words per barrier.
                const uint32_t nn = have - 624u + t;
                auto twist = [](uint32_t hi, uint32_t lo) { const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu); return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); };
                const uint32_t a0 = ring[nn & (kMtRing - 1u)], a1 = ring[(nn + 1u) & (kMtRing - 1u)], am = ring[(nn + 397u) & (kMtRing - 1u)];
                const uint32_t b0 = ring[(nn + 227u) & (kMtRing - 1u)], b1 = ring[(nn + 228u) & (kMtRing - 1u)];
                const uint32_t w1 = am ^ twist(a0, a1), w2 = w1 ^ twist(b0, b1);
                ring[(nn + 624u) & (kMtRing - 1u)] = w1;
                ring[(nn + 851u) & (kMtRing - 1u)] = w2;
                if (t < kMtThird) {                             // and a third one while y[nn + 455] is still an old word: t + 455 < 624
                    const uint32_t c0 = ring[(nn + 454u) & (kMtRing - 1u)], c1 = ring[(nn + 455u) & (kMtRing - 1u)];
                    ring[(nn + 1078u) & (kMtRing - 1u)] = w2 ^ twist(c0, c1);
                }
            }
            have += 2u * kMtStep + kMtThird;                    // 623 words per barrier
            __syncthreads();
        }
        // ---- two paths per thread: local indices l and l + 256 ----
        PathPair pp;
        bool valid[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const uint32_t l = r * kMtRound + (uint32_t)half * 256u + t;
            const uint32_t lc = l < npaths ? l : 0u;            // past the group's end: a harmless stand-in
            const uint4 w4 = *reinterpret_cast<const uint4 *>(&ring[(4u * lc) & (kMtRing - 1u)]);   // raw words of this path; tempered here
            const uint32_t a1 = mt_temper_word(w4.x) >> 5, b1 = mt_temper_word(w4.y) >> 6, a2 = mt_temper_word(w4.z) >> 5, b2 = mt_temper_word(w4.w) >> 6;
            const double u1 = ((double)a1 * 67108864.0 + (double)b1) / 9007199254740992.0; // random_sample
            const double u2 = ((double)a2 * 67108864.0 + (double)b2) / 9007199254740992.0;
            const uint32_t run = lc >> ma.log2_s;               // sub-pixel run of the group: pixel * 4 + sub  (gen_data.py:32-36)
            const uint32_t sub = run & 3u, pl = run >> 2;
            uint32_t jj = j0 + pl, ii = i0;
            while (jj >= H) { jj -= H; ++ii; }
            const uint64_t q = q0 + pl;
            valid[half] = l < npaths && q >= fa.pixel_begin && q < pix_end;
            float rox, roy, roz, rdx, rdy, rdz;
            camera_ray(cam, fa.width, fa.height, ii, jj, sub >> 1, sub & 1u, u1, u2, rox, roy, roz, rdx, rdy, rdz);
            if (half == 0) { pp.ox.x = rox; pp.oy.x = roy; pp.oz.x = roz; pp.dx.x = rdx; pp.dy.x = rdy; pp.dz.x = rdz; }
            else { pp.ox.y = rox; pp.oy.y = roy; pp.oz.y = roz; pp.dx.y = rdx; pp.dy.y = rdy; pp.dz.y = rdz; }
        }
        pp.rx = pp.ry = pp.rz = f2{1.0f, 1.0f};
        trace2_ns8<MODE>(sc, tab8, pp, ta, planes);            // full trace of both paths (lanes past the range compute garbage)
        traced += (valid[0] ? ta.depth : 0u) + (valid[1] ? ta.depth : 0u);
        cbuf[t] = pp.rx.x * gain.r; cbuf[kMtRound + t] = pp.ry.x * gain.g; cbuf[2 * kMtRound + t] = pp.rz.x * gain.b; // render.cpp:194-196
        cbuf[256 + t] = pp.rx.y * gain.r; cbuf[kMtRound + 256 + t] = pp.ry.y * gain.g; cbuf[2 * kMtRound + 256 + t] = pp.rz.y * gain.b;
        __syncthreads();
        // ---- np.mean over every run of S samples of the round (data_visualization.py:41-45): 8 lanes per (run, channel) ----
        const uint32_t nruns = kMtRound >> ma.log2_s, lanes_needed = nruns * 24u;
        for (uint32_t base = 0; base < lanes_needed; base += kBlock) {
            const uint32_t tid = base + t;
            const bool on = tid < lanes_needed;
            const uint32_t task = on ? tid >> 3 : 0u, j = tid & 7u;
            const uint32_t run = task / 3u, ch = task - run * 3u;
            const float *a = cbuf + ch * kMtRound + (run << ma.log2_s);
            auto leaf = [&](const float *x, uint32_t n) __attribute__((always_inline)) { // numpy pairwise_sum, 8 <= n <= 128, n % 8 == 0
                float acc = x[j];
                for (uint32_t i8 = 8; i8 < n; i8 += 8) acc = acc + x[i8 + j];
                acc = acc + __shfl_xor(acc, 1, 64);             // ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7))
                acc = acc + __shfl_xor(acc, 2, 64);
                acc = acc + __shfl_xor(acc, 4, 64);
                return acc;
            };
            float res;
            if (S <= 128) res = leaf(a, S);
            else { const float l0 = leaf(a, 128), l1 = leaf(a + 128, 128); res = l0 + l1; }   // pairwise(256) = pairwise(128) + pairwise(128)
            const uint32_t rg = r * nruns + run;                // run of the group
            if (on && j == 0 && rg < kMtGroupPixels * 4u) submean[rg * 3u + ch] = res / (float)S;   // np.mean: float32 sum / count
        }
        __syncthreads();
    }
    // ---- decode_color (data_visualization.py:36-57): four sub-pixel means in float64, / 4, clip, 8-bit by truncation ----
    for (uint32_t k = t; k < kMtGroupPixels * 3u; k += kBlock) {
        const uint32_t pl = k / 3u, ch = k - pl * 3u;
        const uint64_t q = q0 + pl;
        if (q < fa.pixel_begin || q >= pix_end) continue;
        double acc = 0.0;
#pragma unroll
        for (int sq = 0; sq < 4; ++sq) acc = acc + (double)submean[(pl * 4u + (uint32_t)sq) * 3u + ch];
        const double v = acc / 4;
        const double cl = v < 0 ? 0 : (v > 1 ? 1 : v);
        const uint64_t o = q - fa.pixel_begin;
        fa.fb[(uint64_t)ch * fa.pixel_count + o] = (float)cl;
        if (fa.fb_u8) fa.fb_u8[o * 3 + ch] = (uint8_t)(cl * 255);
    }
    count_traced(ta, traced);
}

// ---- the same pipeline for EVERY sample count (round 4) ---------------------------------------------------------------------------
// 78 pixels are 312 S paths = 2 S generator blocks for ANY S, so the group structure above needs no power of two; what does is the
// summation stage (runs of S samples that tile a 512-path round).  This kernel keeps the colours of the last TWO rounds in a 1024-entry
// LDS ring and, after every round, sums the pairwise LEAVES (pt_leaf.h: numpy's np.mean plan of a run of S samples -- S < 8: a plain
// loop; leaves of <= 128 samples with 8 interleaved accumulators and an in-order n % 8 tail; halves combined through a stack) that END in
// that round: a leaf is at most 128 samples long, so all of it is still in the ring.  One 8-lane task per (run, channel); the partial-sum
// stack of the one run that is still open at the end of a round carries over to the next.  S = 1 (the reference's own default,
// src/common.h:4-6), 2, 4, non-powers of two and counts above 256 run through this kernel; {8, ..., 256} keep the kernel above (its sums
// are a fixed 24 lanes per run: C2 22.6 ms).  Same arithmetic, operation for operation, as gen_rays_mt -> render_do_ex -> decode_color.
constexpr uint32_t kMtColRing = 2 * kMtRound;  // colours kept: this round's and the previous one's
constexpr uint32_t kMtOpenSlots = 8;           // runs that can be open or touched in one round when a run has several leaves (S > 128: <= 5)

template <int MODE>
__global__ __launch_bounds__(kBlock, APT_TWO_WAVES) void render_frame_mt_any_kernel(const float *__restrict__ sph, FrameArgs fa, TraceArgs ta, MtFrameArgs ma,
                                                                                    LeafProg lp) {
    __shared__ float4 tab[kTab8Floats4];
    __shared__ Camera cam;
    __shared__ __align__(16) uint32_t ring[kMtRing];          // y[n] at ring[n % 4096]
    __shared__ float cring[3 * kMtColRing];                    // [channel][path index mod 1024]
    __shared__ float submean[kMtGroupPixels * 4 * 3];          // [pixel of the group][sub-pixel][channel]
    __shared__ float stk[kMtOpenSlots * 3 * kMaxStack];        // pairwise stacks of the runs touched in a round (nleaves > 1 only)
    const uint32_t t = threadIdx.x;
    if (t < sizeof(Camera) / sizeof(double)) (&cam.pos[0])[t] = (&fa.cam.pos[0])[t];
    Scene8 sc;
    const Tab8 tab8 = load_scene8(sph, sc, tab);                // ends with a barrier
    const bool planes = sc.planes;
    const Gain3 gain = load_gain(sph, ta);
    const uint32_t S = fa.samples, H = fa.height, nleaves = lp.nleaves;
    const uint32_t blk = xcd_chunked_block<2>(blockIdx.x, gridDim.x);   // (XCD-aware: neighbouring pixel groups through one L2)
    const uint64_t group = ma.first_group + blk;
    const uint32_t npaths = 312u * S;                           // of this group
    const uint64_t q0 = group * kMtGroupPixels;                 // first pixel
    const uint32_t i0 = (uint32_t)(q0 / H), j0 = (uint32_t)(q0 % H);
    const uint64_t pix_end = fa.pixel_begin + fa.pixel_count;
    for (uint32_t i = t; i < 624; i += kBlock) ring[i] = ma.checkpoints[(uint64_t)blk * 624 + i];   // y[0 .. 623] = the state of the group's first block
    __syncthreads();

    uint32_t have = 624;                                        // words y[0 .. have) exist
    uint32_t traced = 0;
    uint32_t done_run = 0, done_leaf = 0;                       // the first leaf not summed yet: leaf `done_leaf` of run `done_run` (uniform)
    const uint32_t nrounds = (npaths + kMtRound - 1) / kMtRound;
    for (uint32_t r = 0; r < nrounds; ++r) {
        const uint32_t need = min(4u * npaths, 2048u * (r + 1u));
        while (have < need) {                                   // (the generator: as in render_frame_mt_kernel)
            if (t < kMtStep) {
                const uint32_t nn = have - 624u + t;
                auto twist = [](uint32_t hi, uint32_t lo) { const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu); return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); };
                const uint32_t a0 = ring[nn & (kMtRing - 1u)], a1 = ring[(nn + 1u) & (kMtRing - 1u)], am = ring[(nn + 397u) & (kMtRing - 1u)];
                const uint32_t b0 = ring[(nn + 227u) & (kMtRing - 1u)], b1 = ring[(nn + 228u) & (kMtRing - 1u)];
                const uint32_t w1 = am ^ twist(a0, a1), w2 = w1 ^ twist(b0, b1);
                ring[(nn + 624u) & (kMtRing - 1u)] = w1;
                ring[(nn + 851u) & (kMtRing - 1u)] = w2;
                if (t < kMtThird) {
                    const uint32_t c0 = ring[(nn + 454u) & (kMtRing - 1u)], c1 = ring[(nn + 455u) & (kMtRing - 1u)];
                    ring[(nn + 1078u) & (kMtRing - 1u)] = w2 ^ twist(c0, c1);
                }
            }
            have += 2u * kMtStep + kMtThird;
            __syncthreads();
        }
        // ---- two paths per thread: local indices l and l + 256 ----
        PathPair pp;
        bool valid[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const uint32_t l = r * kMtRound + (uint32_t)half * 256u + t;
            const uint32_t lc = l < npaths ? l : 0u;            // past the group's end: a harmless stand-in
            const uint4 w4 = *reinterpret_cast<const uint4 *>(&ring[(4u * lc) & (kMtRing - 1u)]);
            const uint32_t a1 = mt_temper_word(w4.x) >> 5, b1 = mt_temper_word(w4.y) >> 6, a2 = mt_temper_word(w4.z) >> 5, b2 = mt_temper_word(w4.w) >> 6;
            const double u1 = ((double)a1 * 67108864.0 + (double)b1) / 9007199254740992.0; // random_sample
            const double u2 = ((double)a2 * 67108864.0 + (double)b2) / 9007199254740992.0;
            const uint32_t run = lc / S;                        // sub-pixel run of the group: pixel * 4 + sub  (gen_data.py:32-36)
            const uint32_t sub = run & 3u, pl = run >> 2;
            uint32_t jj = j0 + pl, ii = i0;
            while (jj >= H) { jj -= H; ++ii; }
            const uint64_t q = q0 + pl;
            valid[half] = l < npaths && q >= fa.pixel_begin && q < pix_end;
            float rox, roy, roz, rdx, rdy, rdz;
            camera_ray(cam, fa.width, fa.height, ii, jj, sub >> 1, sub & 1u, u1, u2, rox, roy, roz, rdx, rdy, rdz);
            if (half == 0) { pp.ox.x = rox; pp.oy.x = roy; pp.oz.x = roz; pp.dx.x = rdx; pp.dy.x = rdy; pp.dz.x = rdz; }
            else { pp.ox.y = rox; pp.oy.y = roy; pp.oz.y = roz; pp.dx.y = rdx; pp.dy.y = rdy; pp.dz.y = rdz; }
        }
        pp.rx = pp.ry = pp.rz = f2{1.0f, 1.0f};
        trace2_ns8<MODE>(sc, tab8, pp, ta, planes);
        traced += (valid[0] ? ta.depth : 0u) + (valid[1] ? ta.depth : 0u);
        {
            const uint32_t la = (r * kMtRound + t) & (kMtColRing - 1u), lb = (r * kMtRound + 256u + t) & (kMtColRing - 1u);
            cring[la] = pp.rx.x * gain.r; cring[kMtColRing + la] = pp.ry.x * gain.g; cring[2 * kMtColRing + la] = pp.rz.x * gain.b; // render.cpp:194-196
            cring[lb] = pp.rx.y * gain.r; cring[kMtColRing + lb] = pp.ry.y * gain.g; cring[2 * kMtColRing + lb] = pp.rz.y * gain.b;
        }
        __syncthreads();
        // ---- np.mean of every run (data_visualization.py:41-45): the leaves that are complete now, run by run ----
        const uint32_t avail = min(npaths, kMtRound * (r + 1u));   // paths [0, avail) have their colour
        // runs touched: from the open run to the last run with a complete leaf; every task walks its run's leaves in plan order
        const uint32_t first_run = done_run;
        const uint32_t last_run = min(312u - 1u, (avail - 1u) / S);        // the run holding the last available path
        const uint32_t nrun_tasks = last_run - first_run + 1u, lanes_needed = nrun_tasks * 24u;
        for (uint32_t base = 0; base < lanes_needed; base += kBlock) {
            const uint32_t tid = base + t;
            const bool on = tid < lanes_needed;
            const uint32_t task = on ? tid >> 3 : 0u, j = tid & 7u;
            const uint32_t slot = task / 3u, ch = task - slot * 3u;
            const uint32_t run = first_run + slot, rbase = run * S;
            const float *col = cring + ch * kMtColRing;
            float *st = stk + ((slot & (kMtOpenSlots - 1u)) * 3u + ch) * kMaxStack;
            uint32_t start = 0, sp = 0;
            for (uint32_t i = 0; i < nleaves; ++i) {
                const uint32_t n = lp.len(i);
                const bool mine = (run > done_run || i >= done_leaf) && rbase + start + n <= avail;   // not summed yet, and complete
                if (run == done_run && i < done_leaf) { sp += 1u - lp.ncomb(i); start += n; continue; }   // summed in an earlier round (uniform per task)
                if (!mine) break;                               // this leaf (and all later ones) ends in a later round
                // numpy pairwise_sum of a[0 .. n): n < 8 a plain loop from 0; else 8 accumulators, tree, tail in order
                const uint32_t a0 = rbase + start;
                float acc;
                if (n < 8u) {
                    acc = 0.0f;
                    for (uint32_t k = 0; k < n; ++k) acc = acc + col[(a0 + k) & (kMtColRing - 1u)];
                } else {
                    const uint32_t nfull = n & ~7u;
                    acc = col[(a0 + j) & (kMtColRing - 1u)];
                    for (uint32_t i8 = 8; i8 < nfull; i8 += 8) acc = acc + col[(a0 + i8 + j) & (kMtColRing - 1u)];
                    acc = acc + __shfl_xor(acc, 1, 64);         // ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7))
                    acc = acc + __shfl_xor(acc, 2, 64);
                    acc = acc + __shfl_xor(acc, 4, 64);
                    for (uint32_t k = nfull; k < n; ++k) acc = acc + col[(a0 + k) & (kMtColRing - 1u)];   // res += a[i], in order (every lane the same)
                }
                if (nleaves == 1u) {
                    if (on && j == 0) submean[run * 3u + ch] = acc / (float)S;   // np.mean: float32 sum / count
                } else {                                        // pairwise(left) + pairwise(right), innermost first (the 8 lanes hold equal values)
                    float top = acc;
                    uint32_t sp1 = sp + 1u;
                    for (uint32_t m = 0; m < lp.ncomb(i); ++m) { --sp1; top = st[sp1 - 1u] + top; }
                    if (on && j == 0) st[sp1 - 1u] = top;
                    if (i + 1u == nleaves && on && j == 0) submean[run * 3u + ch] = top / (float)S;
                }
                sp += 1u - lp.ncomb(i);
                start += n;
            }
        }
        __syncthreads();
        {   // advance the cursor (uniform): leaves complete up to `avail`
            uint32_t run = done_run, leaf = done_leaf, start = 0;
            for (uint32_t i = 0; i < leaf; ++i) start += lp.len(i);
            while (run < 312u && run * S + start + lp.len(leaf) <= avail) {
                start += lp.len(leaf);
                if (++leaf == nleaves) { leaf = 0; start = 0; ++run; }
            }
            // the open run becomes slot 0 of the next round: move its stack there
            if (nleaves > 1u && run < 312u && run != first_run && t < 3u * kMaxStack) {
                const uint32_t from = (run - first_run) & (kMtOpenSlots - 1u);
                stk[t] = stk[from * 3u * kMaxStack + t];
            }
            done_run = run; done_leaf = leaf;
        }
        __syncthreads();
    }
    // ---- decode_color (data_visualization.py:36-57): four sub-pixel means in float64, / 4, clip, 8-bit by truncation ----
    for (uint32_t k = t; k < kMtGroupPixels * 3u; k += kBlock) {
        const uint32_t pl = k / 3u, ch = k - pl * 3u;
        const uint64_t q = q0 + pl;
        if (q < fa.pixel_begin || q >= pix_end) continue;
        double acc = 0.0;
#pragma unroll
        for (int sq = 0; sq < 4; ++sq) acc = acc + (double)submean[(pl * 4u + (uint32_t)sq) * 3u + ch];
        const double v = acc / 4;
        const double cl = v < 0 ? 0 : (v > 1 ? 1 : v);
        const uint64_t o = q - fa.pixel_begin;
        fa.fb[(uint64_t)ch * fa.pixel_count + o] = (float)cl;
        if (fa.fb_u8) fa.fb_u8[o * 3 + ch] = (uint8_t)(cl * 255);
    }
    count_traced(ta, traced);
}

} // namespace
