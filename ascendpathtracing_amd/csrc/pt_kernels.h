// pt_kernels.h -- the __global__ kernels of librender_mi355x.so (included by render_kernels.hip only):
// buffer-mode render (+ its compaction variant), fused frame, first-hit mode, device gen_rays (counter
// RNG and the reference's MT19937 stream), device decode_color, and the arithmetic self-tests.
#pragma once
#include "pt_trace.h"
#include "pt_trace2.h"
#include "pt_queue.h"
#include "pt_frame_mt.h"

namespace {

// ---- kernel: rays from a buffer ---------------------------------------------------------
template <int MODE, int SC, bool RETIRE>
__global__ __launch_bounds__(kBlock, SC == kSceneGrid ? APT_GRID_WAVES : 1) void render_paths_kernel(const float *__restrict__ rays,
                                                              const float *__restrict__ sph,
                                                              float *__restrict__ colors, uint64_t n_total,
                                                              uint64_t begin, uint64_t count, TraceArgs ta) {
    constexpr bool NS8 = SC == kScene8;
    __shared__ float4 tab[kTab8Floats4];
    __shared__ float4 tile[SC == kSceneTiles ? kTile : 1];
    Scene8 sc;
    Tab8 tab8{tab, tab + 8};
    if (NS8) tab8 = load_scene8(sph, sc, tab);
    const uint64_t local = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool valid = local < count;
    const uint64_t p = begin + (valid ? local : 0);
    PathState s;                                          // CopyIn: render.cpp:82-101
    path_init(s, rays[p], rays[n_total + p], rays[2 * n_total + p], rays[3 * n_total + p], rays[4 * n_total + p],
              rays[5 * n_total + p]);
    uint32_t traced;
    if (SC == kScene8) traced = trace_ns8<MODE, RETIRE>(sc, tab8, s, valid, ta, p);
    else if (SC == kSceneGrid) traced = trace_grid<MODE, RETIRE>(sph, ta.grid, s, valid, ta, p);
    else traced = trace_dyn<MODE, RETIRE>(sph, tile, s, valid, ta, p);
    if (valid) {                                          // render.cpp:194-196, CopyOut :210-223
        const Gain3 gain = load_gain(sph, ta);
        colors[p] = s.rxy.x * gain.r;
        colors[n_total + p] = s.rxy.y * gain.g;
        colors[2 * n_total + p] = s.rz * gain.b;
    }
    count_traced(ta, valid ? traced : 0);
}

// ---- kernel: rays from a buffer, TWO paths per lane (the reference scene, every segment traced, no roulette; large ranges) ----------
// The drop-in boundary at scale (render_do_ex on a C2-sized buffer: the three-kernel form of the reference's exact pipeline): the bounce of
// the fused frame kernel's headline form (pt_trace2.h: every register pair holds (A, B) of one quantity, the whole shading step packed,
// 145.5 issue slots per path and bounce).  A block takes kPaths2Pairs runs of 2 * kBlock consecutive paths, thread t the paths t and t + kBlock
// of a run: both halves load and store coalesced.  Lanes past the end of the range trace a copy of a valid path and store nothing.
constexpr int kPaths2Pairs = 2;   // path pairs a thread of render_paths2_kernel traces one after the other (the block's set-up -- scene to SGPRs and LDS, key constants -- is ~6 % of one pair's 8 bounces)
template <int MODE>
__global__ __launch_bounds__(kBlock, APT_TWO_WAVES) void render_paths2_kernel(const float *__restrict__ rays, const float *__restrict__ sph,
                                                                              float *__restrict__ colors, uint64_t n_total, uint64_t begin,
                                                                              uint64_t count, TraceArgs ta) {
    __shared__ float4 tab[kTab8Floats4];
    Scene8 sc;
    const Tab8 tab8 = load_scene8(sph, sc, tab);           // (with the (1,1,1) entry the two-path bounce multiplies a finished path's throughput by)
    const Gain3 gain = load_gain(sph, ta);                 // render.cpp:194-196
    uint32_t traced = 0;
    for (int it = 0; it < kPaths2Pairs; ++it) {
        const uint64_t local = ((uint64_t)blockIdx.x * kPaths2Pairs + it) * (2 * kBlock) + threadIdx.x;
        if (__builtin_amdgcn_readfirstlane((int)(((uint64_t)blockIdx.x * kPaths2Pairs + it) * (2 * kBlock) >= count))) break;   // (wave-uniform: nothing left for this block)
        const bool va = local < count, vb = local + kBlock < count;
        const uint64_t pa = begin + (va ? local : 0), pb = vb ? begin + local + kBlock : pa;
        PathPair pp;                                       // CopyIn: render.cpp:82-101
        pp.ox = f2{rays[pa], rays[pb]}; pp.oy = f2{rays[n_total + pa], rays[n_total + pb]}; pp.oz = f2{rays[2 * n_total + pa], rays[2 * n_total + pb]};
        pp.dx = f2{rays[3 * n_total + pa], rays[3 * n_total + pb]}; pp.dy = f2{rays[4 * n_total + pa], rays[4 * n_total + pb]};
        pp.dz = f2{rays[5 * n_total + pa], rays[5 * n_total + pb]};
        pp.rx = pp.ry = pp.rz = f2{1.0f, 1.0f};            // render.cpp:116-121
        trace2_ns8<MODE>(sc, tab8, pp, ta, sc.planes);
        if (va) { colors[pa] = pp.rx.x * gain.r; colors[n_total + pa] = pp.ry.x * gain.g; colors[2 * n_total + pa] = pp.rz.x * gain.b; }   // CopyOut :210-223
        if (vb) { colors[pb] = pp.rx.y * gain.r; colors[n_total + pb] = pp.ry.y * gain.g; colors[2 * n_total + pb] = pp.rz.y * gain.b; }
        traced += ((va ? 1u : 0u) + (vb ? 1u : 0u)) * ta.depth;
    }
    count_traced(ta, traced);
}

// ---- kernel: rays from a buffer, with active-ray compaction (APT_FLAG_RETIRE, Ns == 8) ------------
// Buffer mode has no ordering constraint on its outputs (colour p is stored to colors[p]), so the
// wave-level queue is simple: every wave owns kQueueChunk consecutive paths; a lane whose path is
// finished (alive bit cleared, throughput zero, depth reached) takes the next unissued path of the
// chunk -- ballot of the idle lanes, mbcnt prefix rank, p = next + rank -- and loads its ray.
constexpr uint32_t kQueueChunk = 64 * 16;
template <int MODE>
__global__ __launch_bounds__(kBlock) void render_paths_queue_kernel(const float *__restrict__ rays,
                                                                    const float *__restrict__ sph,
                                                                    float *__restrict__ colors, uint64_t n_total,
                                                                    uint64_t begin, uint64_t count, TraceArgs ta) {
    __shared__ float4 tab[kTab8Floats4];
    Scene8 sc;
    const Tab8 tab8 = load_scene8(sph, sc, tab);
    const Gain3 gain = load_gain(sph, ta);
    const KeyConsts kc = make_key_consts(ta.eps);
    // readfirstlane: the wave index is uniform, and the compiler has to know it (the queue's masks live in SGPRs)
    const uint64_t wave = (uint64_t)blockIdx.x * (kBlock / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint64_t next = wave * kQueueChunk;                         // wave-uniform
    const uint64_t end = min(count, next + kQueueChunk);
    uint32_t depth_left = 0, traced = 0;
    uint64_t cur = 0, cur_key = 0;
    // Same loop shape as the frame kernel's queue (see there): refill / bounce / refill / bounce with the state registers
    // exchanging roles, throughput and alive mask updated in place, no merge with the exact form inside the loop.
    PathState s, n;
    path_init(s, 0.f, 0.f, 0.f, 0.f, 0.f, 1.f);
    n = s;
    f2 thr_xy = {1.0f, 1.0f};
    float thr_z = 1.0f;
    uint64_t alive = 0;
    auto refill = [&](PathState &st) -> bool { // idle lanes take the next unissued paths of the chunk; true: chunk finished
        for (;;) {
            const bool want = depth_left == 0;
            const unsigned long long wants = __ballot(want);
            if (next < end && wants) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(wants >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)wants, 0u));
                const uint64_t remaining = end - next;
                const bool take = want && rank < remaining;
                if (take) {
                    cur = begin + next + rank;
                    st.oxy = f2{rays[cur], rays[n_total + cur]}; st.oz = rays[2 * n_total + cur];
                    st.dxy = f2{rays[3 * n_total + cur], rays[4 * n_total + cur]}; st.dz = rays[5 * n_total + cur];
                    thr_xy = f2{1.0f, 1.0f}; thr_z = 1.0f;
                    depth_left = ta.depth;
                    if (ta.rr_start) cur_key = rr_path_key(ta.seed, cur);
                    if (ta.depth == 0) { colors[cur] = gain.r; colors[n_total + cur] = gain.g; colors[2 * n_total + cur] = gain.b; }
                }
                alive |= __builtin_amdgcn_ballot_w64(take);
                next += min((uint64_t)__popcll(wants), remaining);
            }
            if (__any(depth_left != 0)) return false;
            if (next >= end) return true;
        }
    };
    auto post = [&](bool active) {
        if (ta.rr_start && ta.depth - depth_left + 1 >= ta.rr_start) {
            PathState t;
            t.rxy = thr_xy; t.rz = thr_z; t.alive = select_const(alive, 1);
            russian_roulette(t, cur_key, ta.depth - depth_left);
            thr_xy = t.rxy; thr_z = t.rz;
        }
        traced += active ? 1u : 0u;
        depth_left -= active ? 1u : 0u;
        const bool fin = select_const(alive, 1) == 0 || (thr_xy.x == 0.0f && thr_xy.y == 0.0f && thr_z == 0.0f);
        if (active && (depth_left == 0 || fin)) {
            depth_left = 0;
            colors[cur] = thr_xy.x * gain.r;
            colors[n_total + cur] = thr_xy.y * gain.g;
            colors[2 * n_total + cur] = thr_z * gain.b;
        }
    };
    auto step_fast = [&](const PathState &in, PathState &out) -> bool { // true: go exact from `in`, nothing committed
        const bool active = depth_left != 0;
        Albedo albedo;
        uint64_t alive_out = alive;
        const uint64_t redo = bounce_ns8_v2p<MODE>(sc, tab8, in, out, ta, kc, alive_out, albedo, sc.planes) & __builtin_amdgcn_ballot_w64(active);
        if (__builtin_expect(redo != 0, 0)) {
            const bool fin = select_const(alive, 1) == 0 || (thr_xy.x == 0.0f && thr_xy.y == 0.0f && thr_z == 0.0f);
            if (__any(select_const(redo, 1) != 0 && !fin)) return true;
        }
        apply_albedo(thr_xy, thr_z, albedo, alive_out);
        alive = alive_out;
        post(active);
        return false;
    };
    auto step_exact = [&](PathState &in, PathState &out) {
        const bool active = depth_left != 0;
        in.rxy = thr_xy; in.rz = thr_z; in.alive = select_const(alive, 1);
        bounce_ns8_exact<MODE>(sc, tab8, in, out, ta);
        thr_xy = out.rxy; thr_z = out.rz;
        alive = __builtin_amdgcn_ballot_w64(out.alive != 0);
        post(active);
    };
    bool done = refill(s), exact = !eps_allows_rootkey(ta.eps);
    if (!done && !exact) {
        for (;;) {
            if (__builtin_expect(step_fast(s, n), 0)) { exact = true; break; }
            if ((done = refill(n))) break;
            if (__builtin_expect(step_fast(n, s), 0)) { exact = true; s = n; break; }
            if ((done = refill(s))) break;
        }
    }
    if (!done && exact) {
        for (;;) {
            step_exact(s, n);
            s = n;
            if (refill(s)) break;
        }
    }
    count_traced(ta, traced);
}

// ---- kernel: fused frame ----------------------------------------------------------------
// FrameArgs: pt_queue.h

// GROUP lanes share one sub-pixel: lane j of the group owns numpy's pairwise accumulator
// r[j] (samples j, 8+j, 16+j, ...), so the summation order of np.mean is reproduced with
// a 3-step butterfly and no shared memory.  GROUP == 1 serves samples < 8 (numpy sums
// those sequentially).
// TWO: two samples of a lane's chain are traced at a time (pt_trace2.h); only with SC == kScene8, GROUP == 8, no
// retirement, no roulette (the host picks the kernel).
template <int MODE, int SC, int GROUP, bool RETIRE, bool TWO = false>
__global__ __launch_bounds__(kBlock, TWO ? APT_TWO_WAVES : (SC == kSceneGrid ? APT_GRID_WAVES : (SC == kSceneTiles ? APT_TILE_WAVES : APT_FULL_WAVES))) void render_frame_kernel(const float *__restrict__ sph, FrameArgs fa,
                                                              TraceArgs ta, LeafProg lp) {
    constexpr bool NS8 = SC == kScene8;
    static_assert(!(RETIRE && NS8 && GROUP == 8), "APT_FLAG_RETIRE on the 8-sphere scene with samples >= 8 is render_frame_queue8_kernel's (pt_queue.h)");
    if (SC == kSceneGrid && ta.grid_walk == 2u && grid_queue_usable(ta)) return;   // the sample-queue kernel's grid form renders this frame (pt_queue.h)
    __shared__ float4 tab[kTab8Floats4];
    __shared__ float4 tile[SC == kSceneTiles ? kTile : 1];
    extern __shared__ float dyn_lds[];
    float *stack_lds = dyn_lds;                                            // [kMaxStack][3][kStackSlots] when lp.nleaves > 1
    float *queue_lds = dyn_lds + (lp.nleaves > 1 ? kMaxStack * 3 * kStackSlots : 0); // [waves][3][8*maxleaf] (refill)
    // The camera frame (14 doubles) is only needed by ray-generate; parked in LDS it does not
    // occupy 28 SGPRs across the bounce loop (they spilled to VGPR lanes otherwise).
    __shared__ Camera cam;
    if (threadIdx.x < sizeof(Camera) / sizeof(double)) (&cam.pos[0])[threadIdx.x] = (&fa.cam.pos[0])[threadIdx.x];
    Scene8 sc;
    Tab8 tab8{tab, tab + 8};
    if (NS8) tab8 = load_scene8(sph, sc, tab);
    else __syncthreads();
    const bool planes = NS8 && sc.planes;   // wave-uniform: the reference scene's axis-aligned walls (pt_trace.h)
    (void)planes;

    const uint32_t lane = threadIdx.x & 63;
    const uint64_t L = (uint64_t)xcd_chunked_block<16>(blockIdx.x, gridDim.x) * kBlock + threadIdx.x;
    const uint32_t j = (GROUP == 8) ? (uint32_t)(L & 7) : 0u;
    const uint32_t sub = (uint32_t)(L / GROUP) & 3u;
    const uint64_t pl = L / (4 * GROUP);
    const bool valid = pl < fa.pixel_count;
    const uint64_t q = fa.pixel_begin + (valid ? pl : 0);
    const uint32_t pi = (uint32_t)(q / fa.height), pj = (uint32_t)(q % fa.height);
    const uint32_t sy = sub >> 1, sx = sub & 1;
    const uint64_t pbase = (q * 4 + sub) * fa.samples;
    uint32_t traced = 0;       // segments of this lane's own pixel (gated by `valid` at the end)
    uint32_t queue_traced = 0; // refill queue: segments this lane traced for ANY valid item of its wave

    const Gain3 gain = load_gain(sph, ta);
    struct Col { float r, g, b; };
    auto sample = [&](uint32_t k) __attribute__((always_inline)) -> Col {
        double u1, u2;
        path_uniforms(fa.seed, pbase + k, u1, u2);
        float rox, roy, roz, rdx, rdy, rdz;
        camera_ray(cam, fa.width, fa.height, pi, pj, sy, sx, u1, u2, rox, roy, roz, rdx, rdy, rdz);
        PathState s;
        path_init(s, rox, roy, roz, rdx, rdy, rdz);
        if (SC == kScene8) traced += trace_ns8<MODE, RETIRE>(sc, tab8, s, valid, ta, pbase + k);
        else if (SC == kSceneGrid) traced += trace_grid<MODE, RETIRE>(sph, ta.grid, s, valid, ta, pbase + k);
        else traced += trace_dyn<MODE, RETIRE>(sph, tile, s, valid, ta, pbase + k);
        return Col{s.rxy.x * gain.r, s.rxy.y * gain.g, s.rz * gain.b};
    };
    auto add = [](const Col &a, const Col &b) { return Col{a.r + b.r, a.g + b.g, a.b + b.b}; };
    struct Col2 { Col a, b; };
    auto sample2 = [&](uint32_t ka, uint32_t kb) __attribute__((always_inline)) -> Col2 { // samples ka and kb of this lane's sub-pixel, traced together
        PathPair pp;
        {
            double u1, u2;
            float rox, roy, roz, rdx, rdy, rdz;
            path_uniforms(fa.seed, pbase + ka, u1, u2);
            camera_ray(cam, fa.width, fa.height, pi, pj, sy, sx, u1, u2, rox, roy, roz, rdx, rdy, rdz);
            pp.ox.x = rox; pp.oy.x = roy; pp.oz.x = roz; pp.dx.x = rdx; pp.dy.x = rdy; pp.dz.x = rdz;
            path_uniforms(fa.seed, pbase + kb, u1, u2);
            camera_ray(cam, fa.width, fa.height, pi, pj, sy, sx, u1, u2, rox, roy, roz, rdx, rdy, rdz);
            pp.ox.y = rox; pp.oy.y = roy; pp.oz.y = roz; pp.dx.y = rdx; pp.dy.y = rdy; pp.dz.y = rdz;
        }
        pp.rx = pp.ry = pp.rz = f2{1.0f, 1.0f};
        trace2_ns8<MODE>(sc, tab8, pp, ta, planes);
        traced += 2 * ta.depth;
        return Col2{Col{pp.rx.x * gain.r, pp.ry.x * gain.g, pp.rz.x * gain.b}, Col{pp.rx.y * gain.r, pp.ry.y * gain.g, pp.rz.y * gain.b}};
    };

    float res[3] = {0.0f, 0.0f, 0.0f};
    uint32_t start = 0;
    int sp = 0;
    for (uint32_t leaf = 0; leaf < lp.nleaves; ++leaf) {
        const uint32_t n = lp.len(leaf);
        float acc[3];
        if (GROUP == 1) { // n < 8: res = 0; res += a[i]
            Col a = {0.0f, 0.0f, 0.0f};
            for (uint32_t k = 0; k < n; ++k) a = add(a, sample(start + k));
            acc[0] = a.r; acc[1] = a.g; acc[2] = a.b;
        } else {          // 8 <= n <= 128: r[j] chains, tree, tail
            const uint32_t nfull = n & ~7u;
            if (RETIRE && SC == kSceneTiles) {
                // Active-ray compaction with a wave-level work queue for scenes traversed by brute force over LDS tiles, where a
                // "bounce" is one workgroup-synchronous pass over the scene.  (The 8-sphere scene and scenes behind a grid have their own
                // kernel for this, pt_queue.h; round 2's form of this queue for the 8-sphere scene lived here until round 4.)  The 8
                // sub-pixel groups of the wave have 8*nfull samples in this leaf; instead of binding sample k of group g to
                // lane (g, k mod 8), any lane that runs out of work takes the next unissued sample:
                // a ballot of the lanes with an empty one-ray slot, a prefix count (mbcnt) as the
                // rank inside the batch, item = next + rank.  Finished colours are parked in a
                // per-wave LDS array indexed by the sample, and lane (g, j) then adds its own chain
                // j, 8+j, ... from there IN ORDER, so numpy's summation order is untouched and the
                // frame stays bit-identical.  Ray-generate (float64, the expensive part) runs for
                // the whole wave only when >= kRefillLanes lanes want a ray or nothing else is left.
                float *colq = queue_lds + (size_t)(threadIdx.x >> 6) * 3u * 8u * lp.maxleaf; // [3][8*maxleaf]
                const uint32_t total = 8u * nfull;        // items of this wave in this leaf (uniform)
                const uint32_t qstride = 8u * lp.maxleaf;
                uint32_t next = 0;                        // first unissued item (uniform)
                uint32_t depth_left = 0, cur_item = 0, slot_item = 0;
                uint64_t cur_key = 0, slot_key = 0;       // Russian-roulette keys of the running / waiting path
                uint32_t n_bounce_exec = 0, n_gen_exec = 0; // wave-level executions (statistics only)
                float sl_ox = 0.f, sl_oy = 0.f, sl_oz = 0.f, sl_dx = 0.f, sl_dy = 0.f, sl_dz = 1.f; // the one-ray slot
                bool slot_full = false, slot_valid = false, cur_valid = false;
                PathState s;
                path_init(s, 0.f, 0.f, 0.f, 0.f, 0.f, 1.f);
                // Take unissued samples into the one-ray slots (wave-wide float64 ray-generate when enough lanes want one),
                // start waiting rays in `cur`; -> true when the wave has nothing left to do in this leaf.
                auto refill = [&](PathState &cur) -> bool {
                    for (;;) {
                        const bool want = !slot_full;
                        const unsigned long long wants = __ballot(want);
                        const bool busy_any = __any(depth_left != 0 || slot_full);
                        if (next < total && wants && ((uint32_t)__popcll(wants) >= ta.refill_lanes || !busy_any)) {
                            ++n_gen_exec;
                            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(wants >> 32),
                                                  __builtin_amdgcn_mbcnt_lo((uint32_t)wants, 0u));
                            const uint32_t remaining = total - next;
                            const bool take = want && rank < remaining;
                            // group coordinates come from the first lane of the item's group; every lane
                            // of the wave takes part in the shuffles (a masked-off source lane would
                            // return garbage), lanes that do not take an item use item 0
                            const uint32_t item = take ? next + rank : 0u;
                            const uint32_t g = item / nfull, k = item - g * nfull;
                            const int src = (int)(8u * g);
                            const uint32_t gpi = __shfl(pi, src, 64), gpj = __shfl(pj, src, 64);
                            const uint32_t gsub = __shfl(sub, src, 64);
                            const uint32_t blo = __shfl((uint32_t)pbase, src, 64), bhi = __shfl((uint32_t)(pbase >> 32), src, 64);
                            const bool gvalid = __shfl((int)valid, src, 64) != 0;
                            if (take) {
                                slot_valid = gvalid;
                                double u1, u2;
                                const uint64_t path = (((uint64_t)bhi << 32) | blo) + start + k;
                                if (ta.rr_start) slot_key = rr_path_key(ta.seed, path);
                                path_uniforms(fa.seed, path, u1, u2);
                                camera_ray(cam, fa.width, fa.height, gpi, gpj, gsub >> 1, gsub & 1u, u1, u2, sl_ox, sl_oy, sl_oz, sl_dx,
                                           sl_dy, sl_dz);
                                slot_item = item;
                                slot_full = true;
                            }
                            next += min((uint32_t)__popcll(wants), remaining);
                        }
                        const bool begin = depth_left == 0 && slot_full; // start the waiting ray
                        if (begin) {
                            cur.oxy = f2{sl_ox, sl_oy}; cur.oz = sl_oz; cur.dxy = f2{sl_dx, sl_dy}; cur.dz = sl_dz;
                            cur.rxy = f2{1.0f, 1.0f}; cur.rz = 1.0f; cur.alive = 1u;
                            cur_item = slot_item;
                            cur_key = slot_key;
                            cur_valid = slot_valid;
                            depth_left = ta.depth;
                            slot_full = false;
                            if (ta.depth == 0 || !cur_valid) { // depth 0, or a group past the image: colour = gain
                                depth_left = 0;
                                colq[cur_item] = gain.r; colq[qstride + cur_item] = gain.g; colq[2 * qstride + cur_item] = gain.b;
                            }
                        }
                        if (__any(depth_left != 0)) return false;
                        if (next >= total && !__any(slot_full)) return true;
                    }
                };
                // The scene pass contains workgroup barriers, so the four waves of the workgroup take their passes together and
                // leave together; a wave whose queue is empty keeps staging tiles with all its lanes inactive.  Every lane does
                // the same work per segment here, so what the queue buys is exactly the dead lane-segments (19 % at depth 8 on
                // the 10 000-sphere scene, most of them with roulette).
                for (;;) {
                    const bool wave_done = refill(s);
                    if (__syncthreads_and(wave_done)) break;
                    const bool active = depth_left != 0;
                    dyn_segment<MODE>(sph, tile, s, !active, ta);
                    ++n_bounce_exec;
                    if (active && ta.rr_start && ta.depth - depth_left + 1 >= ta.rr_start)
                        russian_roulette(s, cur_key, ta.depth - depth_left);
                    queue_traced += (active && cur_valid) ? 1u : 0u; // the item's pixel decides, not this lane's own
                    depth_left -= active ? 1u : 0u;
                    if (active && (depth_left == 0 || path_finished(s))) {
                        depth_left = 0;
                        colq[cur_item] = s.rxy.x * gain.r;
                        colq[qstride + cur_item] = s.rxy.y * gain.g;
                        colq[2 * qstride + cur_item] = s.rz * gain.b;
                    }
                }
                __syncthreads(); // colours of the whole leaf are in LDS (only wave-local data is read back)
                {   // lane (g, j) adds samples j, 8+j, ... of its own group, in order: numpy's r[j] chain
                    const uint32_t base = (lane >> 3) * nfull + j;
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) acc[ch] = colq[ch * qstride + base];
                    for (uint32_t i8 = 8; i8 < nfull; i8 += 8) {
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) acc[ch] = acc[ch] + colq[ch * qstride + base + i8];
                    }
                }
                __syncthreads(); // before the next leaf reuses the array
                if (ta.traced && lane == 0) { // lane-slots spent: executions x 64
                    atomicAdd(ta.traced + 1, 64ull * n_bounce_exec);
                    atomicAdd(ta.traced + 2, 64ull * n_gen_exec);
                }
            } else {
                Col a;
                uint32_t i8;
                if (TWO && nfull >= 16) { // numpy's chain r[j] += a[j + 8m], two members at a time, added in order
                    Col2 c = sample2(start + j, start + 8 + j);
                    a = add(c.a, c.b);
                    for (i8 = 16; i8 + 16 <= nfull; i8 += 16) {
                        c = sample2(start + i8 + j, start + i8 + 8 + j);
                        a = add(a, c.a);
                        a = add(a, c.b);
                    }
                } else {
                    a = sample(start + j);
                    i8 = 8;
                }
                for (; i8 < nfull; i8 += 8) a = add(a, sample(start + i8 + j));
                acc[0] = a.r; acc[1] = a.g; acc[2] = a.b;
            }
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) { // ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7))
                float v = acc[ch];
                v = v + __shfl_xor(v, 1, 64);
                v = v + __shfl_xor(v, 2, 64);
                v = v + __shfl_xor(v, 4, 64);
                acc[ch] = v;
            }
            const uint32_t nt = n - nfull;
            if (nt) { // res += a[i] for the n % 8 trailing samples, in order
                const uint32_t traced_before = traced;
                const Col c = sample(start + nfull + (j < nt ? j : 0));
                if (j >= nt) traced = traced_before;   // lanes past the tail re-trace its first sample: not a segment of the frame (statistic only)
                for (uint32_t t = 0; t < nt; ++t) {
                    const int src = (int)((lane & ~7u) + t);
                    acc[0] = acc[0] + __shfl(c.r, src, 64);
                    acc[1] = acc[1] + __shfl(c.g, src, 64);
                    acc[2] = acc[2] + __shfl(c.b, src, 64);
                }
            }
        }
        start += n;
        if (lp.nleaves == 1) {
            res[0] = acc[0]; res[1] = acc[1]; res[2] = acc[2];
        } else { // pairwise(left) + pairwise(right), innermost first
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) stack_lds[(sp * 3 + ch) * kStackSlots + (threadIdx.x >> 3)] = acc[ch];
            ++sp;
            for (uint32_t m = 0; m < lp.ncomb(leaf); ++m) {
                --sp;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const float a = stack_lds[((sp - 1) * 3 + ch) * kStackSlots + (threadIdx.x >> 3)];
                    const float b = stack_lds[(sp * 3 + ch) * kStackSlots + (threadIdx.x >> 3)];
                    stack_lds[((sp - 1) * 3 + ch) * kStackSlots + (threadIdx.x >> 3)] = a + b;
                }
            }
        }
    }
    if (lp.nleaves > 1) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) res[ch] = stack_lds[ch * kStackSlots + (threadIdx.x >> 3)];
    }

    // decode_color: data_visualization.py:36-57
    // The 8-bit pixels of a workgroup (kBlock / (4 * GROUP) consecutive pixels, 3 bytes each: a whole number of dwords that starts on a
    // dword when the image does) leave as DWORD stores assembled in LDS instead of three byte stores per pixel from different waves: no partial
    // dwords for the L2 to merge.  (Measured: what brought the frame's write traffic to exactly its size was the XCD-aware block mapping,
    // pt_trace.h; this took the launch's fetched bytes from 0.88 to 0.83 MB.  Kept: it costs one barrier per workgroup.)
    constexpr uint32_t kPixPerBlock = kBlock / (4 * GROUP), kU8Words = kPixPerBlock * 3 / 4;
    static_assert(kPixPerBlock * 3 % 4 == 0, "a workgroup's 8-bit pixels are whole dwords");
    __shared__ uint32_t u8pack[kU8Words];
    const uint64_t pl0 = pl - (threadIdx.x / (4 * GROUP));                          // first pixel of this workgroup (wave-uniform arithmetic on L)
    const bool pack = fa.fb_u8 && pl0 + kPixPerBlock <= fa.pixel_count && (((uintptr_t)fa.fb_u8 + pl0 * 3) & 3u) == 0;   // workgroup-uniform
    const float fs = (float)fa.samples;
    const int gbase = (int)(lane & ~(uint32_t)(4 * GROUP - 1));
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const float mean = res[ch] / fs;            // np.mean: float32 sum / count
        double acc = 0.0;                           // :38 sum_color = zeros (float64)
#pragma unroll
        for (int sq = 0; sq < 4; ++sq) acc = acc + (double)__shfl(mean, gbase + sq * GROUP, 64); // :41-45
        const double v = acc / 4;                   // :46
        const double cl = v < 0 ? 0 : (v > 1 ? 1 : v); // :54
        if (valid && (lane & (4 * GROUP - 1)) == 0) {
            fa.fb[(uint64_t)ch * fa.pixel_count + pl] = (float)cl;
            const uint8_t b8 = (uint8_t)(cl * 255);                               // :55-57 truncation
            if (pack) reinterpret_cast<uint8_t *>(u8pack)[(threadIdx.x / (4 * GROUP)) * 3 + ch] = b8;
            else if (fa.fb_u8) fa.fb_u8[pl * 3 + ch] = b8;
        }
    }
    if (pack) {                                     // (workgroup-uniform: every thread reaches the barrier)
        __syncthreads();
        if (threadIdx.x < kU8Words) reinterpret_cast<uint32_t *>(fa.fb_u8 + pl0 * 3)[threadIdx.x] = u8pack[threadIdx.x];
    }
    count_traced(ta, (valid ? traced : 0) + queue_traced);
}

// ---- kernel: first-hit debug oracle (gen_data.py:134-188 test_scene) ---------------------------
// out[3][N]: emission of the light when it is the first hit, the sphere's colour otherwise, 0 when
// nothing is hit.  One lane per ray, spheres read straight from the [10][Ns] planes (L2-resident).
__global__ __launch_bounds__(kBlock) void test_scene_kernel(const float *__restrict__ rays,
                                                            const float *__restrict__ sph, float *__restrict__ out,
                                                            uint64_t n_total, uint32_t ns, int32_t light, float eps) {
    const uint64_t p = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= n_total) return;
    const float ox = rays[p], oy = rays[n_total + p], oz = rays[2 * n_total + p];
    const float dx = rays[3 * n_total + p], dy = rays[4 * n_total + p], dz = rays[5 * n_total + p];
    float mind = kMissT;
    int id = -1;
    for (uint32_t k = 0; k < ns; ++k)
        test_scene_sphere(sph[ns + k], sph[2 * (size_t)ns + k], sph[3 * (size_t)ns + k], sph[k], ox, oy, oz, dx, dy, dz,
                          eps, (int)k, mind, id);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = 0.0f;
        if (id >= 0) v = (id == light) ? sph[(size_t)(4 + c) * ns + id] : sph[(size_t)(7 + c) * ns + id]; // :175-180
        out[(uint64_t)c * n_total + p] = v;
    }
}

// ---- kernel: device gen_rays (counter RNG) -----------------------------------------------
__global__ __launch_bounds__(kBlock) void gen_rays_kernel(Camera cam, uint32_t width, uint32_t height,
                                                          uint32_t samples, uint64_t seed, uint64_t n_total,
                                                          uint64_t begin, uint64_t count, float *__restrict__ rays) {
    const uint64_t local = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (local >= count) return;
    const uint64_t p = begin + local;
    uint32_t i, j, sy, sx;
    path_coords(p, height, samples, i, j, sy, sx);
    double u1, u2;
    path_uniforms(seed, p, u1, u2);
    float rox, roy, roz, rdx, rdy, rdz;
    camera_ray(cam, width, height, i, j, sy, sx, u1, u2, rox, roy, roz, rdx, rdy, rdz);
    rays[p] = rox; rays[n_total + p] = roy; rays[2 * n_total + p] = roz;
    rays[3 * n_total + p] = rdx; rays[4 * n_total + p] = rdy; rays[5 * n_total + p] = rdz;
}

// ---- kernel: device gen_rays, bit-exact with the reference's MT19937 stream -------------------
// np.random.rand() takes two MT19937 words per double and gen_rays two doubles per path, in path
// order (gen_data.py:32-40), so output block b of the generator (624 words) is exactly paths
// [156b, 156b+156).  One workgroup per checkpoint: load the raw state of block cb = i*stride into
// LDS, emit that block, then `twist` forward block by block.  The twist is the textbook 3-phase
// parallel form: x[i] depends on x[i], x[i+1] and x[i+397], so [0,227), [227,454), [454,624) can each
// be updated at once (read, barrier, write, barrier).
constexpr int kMtN = 624, kMtM = 397, kPathsPerBlock = 156;

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

constexpr int kMtGroup = 8; // output blocks tempered into LDS before the rays of their 8*156 paths are made

__global__ __launch_bounds__(kBlock) void gen_rays_mt_kernel(const uint32_t *__restrict__ checkpoints, uint32_t stride,
                                                             uint64_t first_block, uint64_t num_blocks, Camera cam, uint32_t width,
                                                             uint32_t height, uint32_t samples, uint64_t n_total,
                                                             uint64_t begin, uint64_t end, float *__restrict__ rays) {
    __shared__ uint32_t mt[kMtN];
    __shared__ __align__(16) uint32_t outw[kMtGroup * kMtN]; // tempered output words of up to kMtGroup consecutive blocks (20 KB); read back as uint4
    const uint64_t cb = first_block + (uint64_t)blockIdx.x * stride; // first output block of this workgroup (checkpoints[0] = block first_block)
    for (int i = threadIdx.x; i < kMtN; i += kBlock) mt[i] = checkpoints[(uint64_t)blockIdx.x * kMtN + i];
    __syncthreads();
    const uint64_t last = min(cb + stride, num_blocks);
    for (uint64_t g0 = cb; g0 < last; g0 += kMtGroup) {
        const uint32_t nb = (uint32_t)min((uint64_t)kMtGroup, last - g0);
        for (uint32_t bl = 0; bl < nb; ++bl) {
            if (g0 + bl != cb) { // twist to the next block
                const int t = threadIdx.x;
                const int lo[3] = {0, 227, 454}, hi[3] = {227, 454, 624};
#pragma unroll
                for (int ph = 0; ph < 3; ++ph) {
                    const int i = lo[ph] + t;
                    uint32_t v = 0;
                    const bool on = i < hi[ph];
                    if (on) {
                        const uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % kMtN] & 0x7fffffffu);
                        v = mt[(i + kMtM) % kMtN] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
                    }
                    __syncthreads();
                    if (on) mt[i] = v;
                    __syncthreads();
                }
            }
            for (int i = threadIdx.x; i < kMtN; i += kBlock) outw[bl * kMtN + i] = mt_temper(mt[i]);
        }
        __syncthreads();
        // The camera maths (float64, the expensive part) runs over the group's nb*156 paths with all 256 threads:
        // one block at a time would keep 156 of them busy.  Path coordinates from the group's base with 32-bit
        // divisions (p itself needs 64 bits; p / samples and the pixel index normally do not).
        const uint64_t pbase = g0 * kPathsPerBlock;
        const uint64_t rb = pbase / samples;
        const uint32_t kb = (uint32_t)(pbase - rb * samples);
        const uint32_t npaths = nb * kPathsPerBlock;
        for (uint32_t q = threadIdx.x; q < npaths; q += kBlock) {
            const uint64_t p = pbase + q;
            if (p < begin || p >= end) continue;
            const uint4 w4 = *reinterpret_cast<const uint4 *>(&outw[4 * q]);
            const uint32_t a1 = w4.x >> 5, b1 = w4.y >> 6, a2 = w4.z >> 5, b2 = w4.w >> 6;
            const double u1 = ((double)a1 * 67108864.0 + (double)b1) / 9007199254740992.0; // random_sample
            const double u2 = ((double)a2 * 67108864.0 + (double)b2) / 9007199254740992.0;
            const uint64_t r = rb + (kb + q) / samples;      // p / samples  (gen_data.py:32-36)
            const uint32_t sx = (uint32_t)(r & 1), sy = (uint32_t)((r >> 1) & 1);
            const uint64_t r4 = r >> 2;
            uint32_t i, j;
            if (r4 <= 0xffffffffull) { i = (uint32_t)r4 / height; j = (uint32_t)r4 - i * height; }
            else { i = (uint32_t)(r4 / height); j = (uint32_t)(r4 % height); }
            float rox, roy, roz, rdx, rdy, rdz;
            camera_ray(cam, width, height, i, j, sy, sx, u1, u2, rox, roy, roz, rdx, rdy, rdz);
            rays[p] = rox; rays[n_total + p] = roy; rays[2 * n_total + p] = roz;
            rays[3 * n_total + p] = rdx; rays[4 * n_total + p] = rdy; rays[5 * n_total + p] = rdz;
        }
        __syncthreads(); // outw is rewritten by the next group
    }
}

// ---- kernel: device decode_color -----------------------------------------------------------
__device__ float pairwise_leaf(const float *a, uint32_t n) { // numpy pairwise_sum, n <= 128
    if (n < 8) {
        float r = 0.0f;
        for (uint32_t i = 0; i < n; ++i) r = r + a[i];
        return r;
    }
    float r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = a[k];
    uint32_t i;
    for (i = 8; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = r[k] + a[i + k];
    }
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res = res + a[i];
    return res;
}

// one thread per (pixel, channel, sub-pixel); 4 adjacent lanes combine in float64
__global__ __launch_bounds__(kBlock) void decode_color_kernel(const float *__restrict__ colors, uint32_t samples,
                                                              uint64_t npix, LeafProg lp, float *__restrict__ fb,
                                                              uint8_t *__restrict__ fb_u8) {
    const uint64_t L = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint32_t sub = (uint32_t)(L & 3);
    const uint64_t pc = L >> 2; // pixel * 3 + channel, channel-major: pc = ch * npix + q
    const bool valid = pc < 3 * npix;
    const uint64_t ch = valid ? pc / npix : 0, q = valid ? pc % npix : 0;
    const uint64_t n_total = npix * 4 * samples;
    const float *a = colors + ch * n_total + (q * 4 + sub) * samples;
    float st[kMaxStack];
    int sp = 0;
    uint32_t start = 0;
    for (uint32_t leaf = 0; leaf < lp.nleaves; ++leaf) {
        st[sp++] = pairwise_leaf(a + start, lp.len(leaf));
        start += lp.len(leaf);
        for (uint32_t m = 0; m < lp.ncomb(leaf); ++m) { --sp; st[sp - 1] = st[sp - 1] + st[sp]; }
    }
    const float mean = st[0] / (float)samples;
    const int gbase = (int)((threadIdx.x & 63) & ~3u);
    double acc = 0.0;
#pragma unroll
    for (int sq = 0; sq < 4; ++sq) acc = acc + (double)__shfl(mean, gbase + sq, 64);
    const double v = acc / 4;
    const double cl = v < 0 ? 0 : (v > 1 ? 1 : v);
    if (valid && sub == 0) {
        fb[ch * npix + q] = (float)cl;
        if (fb_u8) fb_u8[q * 3 + ch] = (uint8_t)(cl * 255);
    }
}

// Same result with coalesced loads (samples >= 8): 8 lanes share one sub-pixel row, lane j owns numpy's accumulator
// r[j] (elements j, 8+j, ...), so one load instruction of the wave reads 8 x 32 contiguous bytes instead of 64
// separate rows; the tree is a 3-step butterfly, the n % 8 tail is added in order through shuffles, leaves above
// 128 samples combine through a per-group LDS stack (the accumulation skeleton of render_frame_kernel).  The 4
// sub-pixel groups of a (pixel, channel) are adjacent in the wave and are summed in float64 in sub-pixel order.
__global__ __launch_bounds__(kBlock) void decode_color_kernel8(const float *__restrict__ colors, uint32_t samples,
                                                               uint64_t npix, LeafProg lp, float *__restrict__ fb,
                                                               uint8_t *__restrict__ fb_u8) {
    __shared__ float stack_lds[kMaxStack * kStackSlots];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t j = threadIdx.x & 7u;
    const uint32_t sub = (threadIdx.x >> 3) & 3u;
    const uint32_t slot = threadIdx.x >> 3;
    const uint64_t n_total = npix * 4 * samples;
    const uint64_t items = 3 * npix, per_block = kBlock / 32;          // (pixel, channel) pairs, channel-major
    const uint64_t rounds = (items + per_block * gridDim.x - 1) / (per_block * gridDim.x);
    // grid-stride over the (pixel, channel) pairs: a block is a few KB of work per round, so the grid is capped and loops (one round at C2
    // would be 777 600 workgroups: bound by the dispatch rate, 5.5 TB/s; the cap the host uses, 256 x 1024, reads 6.1)
    for (uint64_t r = 0; r < rounds; ++r) {
    const uint64_t pc = (r * gridDim.x + blockIdx.x) * per_block + (threadIdx.x >> 5);
    const bool valid = pc < items;
    const uint64_t ch = valid ? pc / npix : 0, q = valid ? pc % npix : 0;
    const float *a = colors + ch * n_total + (q * 4 + sub) * samples;
    float res = 0.0f;
    uint32_t start = 0;
    int sp = 0;
    for (uint32_t leaf = 0; leaf < lp.nleaves; ++leaf) {
        const uint32_t n = lp.len(leaf), nfull = n & ~7u;
        // a leaf has at most 128 samples = 16 per lane: all loads of the lane are issued before the first add
        // (the rolled loop had one 4-byte load in flight per lane and reached 3.9 TB/s)
        float v[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) v[m] = (8u * m < nfull) ? a[start + 8u * m + j] : 0.0f; // wave-uniform guard
        float acc = v[0];
#pragma unroll
        for (int m = 1; m < 16; ++m)
            if (8u * m < nfull) acc = acc + v[m];
        acc = acc + __shfl_xor(acc, 1, 64); // ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7))
        acc = acc + __shfl_xor(acc, 2, 64);
        acc = acc + __shfl_xor(acc, 4, 64);
        const uint32_t nt = n - nfull;
        if (nt) { // res += a[i] for the n % 8 trailing samples, in order
            const float c = a[start + nfull + (j < nt ? j : 0)];
            for (uint32_t t = 0; t < nt; ++t) acc = acc + __shfl(c, (int)((lane & ~7u) + t), 64);
        }
        start += n;
        if (lp.nleaves == 1) {
            res = acc;
        } else { // pairwise(left) + pairwise(right), innermost first; the 8 lanes of a group hold equal values
            stack_lds[sp * kStackSlots + slot] = acc;
            ++sp;
            for (uint32_t m = 0; m < lp.ncomb(leaf); ++m) {
                --sp;
                const float x = stack_lds[(sp - 1) * kStackSlots + slot], y = stack_lds[sp * kStackSlots + slot];
                stack_lds[(sp - 1) * kStackSlots + slot] = x + y;
            }
        }
    }
    if (lp.nleaves > 1) res = stack_lds[slot];
    const float mean = res / (float)samples;              // np.mean: float32 sum / count
    const int gbase = (int)(lane & ~31u);
    double acc64 = 0.0;                                    // data_visualization.py:38 sum_color = zeros (float64)
#pragma unroll
    for (int sq = 0; sq < 4; ++sq) acc64 = acc64 + (double)__shfl(mean, gbase + sq * 8, 64); // :41-45
    const double v = acc64 / 4;                            // :46
    const double cl = v < 0 ? 0 : (v > 1 ? 1 : v);         // :54
    if (valid && (lane & 31u) == 0) {
        fb[ch * npix + q] = (float)cl;
        if (fb_u8) fb_u8[q * 3 + ch] = (uint8_t)(cl * 255); // :55-57 truncation
    }
    }
}

// The same sum with FLOAT4 loads, for small sample counts (8 <= samples <= 32, samples % 4 == 0): 2 lanes per sub-pixel row, lane L owns
// numpy's accumulators r[4L .. 4L+3] and reads a[8m + 4L .. 8m + 4L + 3] as one float4 per block m of 8 samples; 8 lanes = one (pixel,
// channel).  With 8 lanes per row a lane of decode_color_kernel8 has ONE dword per load instruction and 1-4 loads per row to cover the
// latency with: 1.2 / 2.3 TB/s at S = 8 / 16 against 3.2 / 5.0 for this form (profiles/microbench/decode_rates.hip; from S = 64 on the
// 8-lane form with a larger grid is the faster one: 6.1 TB/s, 0.97 of what a float4 copy reaches on this part).  Same additions in the
// same order: ((r0+r1)+(r2+r3)) in lane 0, ((r4+r5)+(r6+r7)) in lane 1, their sum through one shuffle (commutative: both lanes hold the
// same bits), the n % 8 tail in order.
constexpr int kDecode4Groups = kBlock / 2;   // sub-pixel groups (2 lanes) per block
constexpr int kDecode4Blocks = 4;            // blocks of 8 samples a row may have: samples <= 32
__global__ __launch_bounds__(kBlock) void decode_color_kernel4(const float *__restrict__ colors, uint32_t samples, uint64_t npix, LeafProg lp,
                                                               float *__restrict__ fb, uint8_t *__restrict__ fb_u8) {
    __shared__ float stack_lds[kMaxStack * kDecode4Groups];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t L = threadIdx.x & 1u;
    const uint32_t sub = (threadIdx.x >> 1) & 3u;
    const uint32_t slot = threadIdx.x >> 1;
    const uint64_t n_total = npix * 4 * samples;
    const uint64_t items = 3 * npix, per_block = kBlock / 8;           // (pixel, channel) pairs, channel-major
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
            // samples <= 32 (the host's condition for this kernel): ONE leaf of at most 4 blocks of 8, all loads issued before the first add
            // (sixteen float4 in flight -- any leaf -- cost the occupancy more than they hide: 3.2 against 4.7 TB/s at S = 8)
            float4 v[kDecode4Blocks];
#pragma unroll
            for (int m = 0; m < kDecode4Blocks; ++m)                     // wave-uniform guard
                v[m] = (8u * m < nfull) ? *reinterpret_cast<const float4 *>(a + start + 8u * m + 4u * L) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 acc = v[0];
#pragma unroll
            for (int m = 1; m < kDecode4Blocks; ++m)
                if (8u * m < nfull) { acc.x = acc.x + v[m].x; acc.y = acc.y + v[m].y; acc.z = acc.z + v[m].z; acc.w = acc.w + v[m].w; }
            float s = (acc.x + acc.y) + (acc.z + acc.w);                // (r0+r1)+(r2+r3) resp. (r4+r5)+(r6+r7)
            s = s + __shfl_xor(s, 1, 64);
            const uint32_t nt = n - nfull;
            if (nt) {                                                    // res += a[i] for the n % 8 trailing samples, in order
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
            } else {                                                     // pairwise(left) + pairwise(right), innermost first
                stack_lds[sp * kDecode4Groups + slot] = s;
                ++sp;
                for (uint32_t m = 0; m < lp.ncomb(leaf); ++m) {
                    --sp;
                    const float x = stack_lds[(sp - 1) * kDecode4Groups + slot], y = stack_lds[sp * kDecode4Groups + slot];
                    stack_lds[(sp - 1) * kDecode4Groups + slot] = x + y;
                }
            }
        }
        if (lp.nleaves > 1) res = stack_lds[slot];
        const float mean = res / (float)samples;              // np.mean: float32 sum / count
        const int gbase = (int)(lane & ~7u);
        double acc64 = 0.0;                                    // data_visualization.py:38 sum_color = zeros (float64)
#pragma unroll
        for (int sq = 0; sq < 4; ++sq) acc64 = acc64 + (double)__shfl(mean, gbase + sq * 2, 64); // :41-45
        const double v64 = acc64 / 4;                          // :46
        const double cl = v64 < 0 ? 0 : (v64 > 1 ? 1 : v64);   // :54
        if (valid && (lane & 7u) == 0) {
            fb[ch * npix + q] = (float)cl;
            if (fb_u8) fb_u8[q * 3 + ch] = (uint8_t)(cl * 255); // :55-57 truncation
        }
    }
}

// ---- kernel: exhaustive self-test of the fast correctly-rounded sqrt ------------------------
// Every float bit pattern in [begin, begin+count): variant(x) must equal sqrtf(x) bit for bit
// (any NaN == any NaN) unless the variant asks for the fallback (|x| < 2^-96), in which case the
// hot loop would have used sqrtf() anyway.  Counts mismatches; remembers the first one.
__global__ __launch_bounds__(kBlock) void selftest_sqrt_kernel(int variant, uint64_t begin, uint64_t count,
                                                               unsigned long long *result) {
#if defined(__HIP_DEVICE_COMPILE__) // the sqrt variants are device-only builtins
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    unsigned long long bad = 0, first = ~0ull;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += stride) {
        const uint32_t bits = (uint32_t)(begin + i);
        const float x = __uint_as_float(bits);
        float amin = 1.0f;
        const float got = variant == 0 ? sqrt_rn_core(x, amin) : variant == 1 ? sqrt_rn_markstein(x, amin)
                        : variant == 2 ? sqrt_rn_rsq1(x, amin) : sqrt_rn_rsq2(x, amin);
        const float want = sqrtf(x);
        const bool fallback = amin < 0x1p-96f;
        const bool same = (__float_as_uint(got) == __float_as_uint(want)) || (got != got && want != want);
        if (!same && !fallback) { ++bad; if (first == ~0ull) first = bits; }
    }
    if (bad) {
        atomicAdd(&result[0], bad);
        atomicMin(&result[1], first);
    }
#endif
}

// ---- kernel: self-test of the shared-reciprocal divide --------------------------------------
// Operand set i of [begin, begin+count): three numerators built from a counter hash; a quarter of the sets use
// special mantissas (all ones, 1.0, powers of two, one-bit neighbours), exponents at the edges of the accepted
// range and signed zeros.  The divisor is formed in one of three ways, chosen by the counter:
//   0  as the K-mode shading step does: sqrtf of the fp32 sum of squares (rt_helper.h:641-658)
//   1  as the O-mode shading step does: sqrtf of the float64-accumulated sum of float32 squares (gen_data.py:347)
//   2  INDEPENDENT of the numerators: any float in the accepted range [2^-48, 2^30] (len2 := its square, only the
//      validity flags look at it), so the sequence is not only checked on |quotient| <= 1
// Both forms of the sequence (div3_shared with its own flags, div3_packed under div3_operands_ok) must give the
// three quotients of the plain `/` bit for bit on every set their validity test accepts.
__global__ __launch_bounds__(kBlock) void selftest_div3_kernel(uint64_t begin, uint64_t count,
                                                               unsigned long long *result) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    unsigned long long bad = 0, first = ~0ull, accepted = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {   // the validity chain's minimum must PROPAGATE NaN (v_minimum3_f32) and take |.|
        const float qn = __uint_as_float(0x7fc00000u);
        float one = 1.0f, a = -0.5f, b = 3.0f;
        asm volatile("" : "+v"(one), "+v"(a), "+v"(b));
        const float m0 = minimum3_abs(one, a, b), m1 = minimum3_abs(one, qn, b), m2 = minimum3_abs(one, b, -qn), m3 = minimum3_abs(qn, a, b);
        if (!(m0 == 0.5f) || m1 == m1 || m2 == m2 || m3 == m3 || (m1 >= kFastMin) || !(m0 >= kFastMin)) { ++bad; first = 0; }
    }
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += stride) {
        const uint64_t ctr = begin + i;
        uint64_t h = splitmix64(ctr);
        float v[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            h = splitmix64(h);
            // exponents over the whole accepted range and a little beyond it on both sides
            uint32_t man = (uint32_t)h & 0x7fffffu, ex = 127u - 100u + (uint32_t)((h >> 23) % 134u), sg = (uint32_t)(h >> 63);
            if ((ctr & 3u) == 0u) {
                const uint32_t pick = (uint32_t)(h >> 40) & 7u;
                man = pick == 0 ? 0x7fffffu : pick == 1 ? 0u : pick == 2 ? 1u : pick == 3 ? 0x7ffffeu
                    : pick == 4 ? 0x400000u : pick == 5 ? 0x3fffffu : pick == 6 ? 0x400001u : man;
                if (((h >> 44) & 3u) == 0u) ex = ((h >> 46) & 1u) ? 127u - 96u : 127u + 29u;
            }
            v[k] = __uint_as_float((sg << 31) | (ex << 23) | man);
        }
        if ((ctr & 63u) == 1u) v[(ctr >> 6) % 3u] = (ctr & 64u) ? 0.0f : -0.0f; // zero numerators
        const uint32_t how = (uint32_t)((ctr >> 2) % 3u);
        float len2, d;
        if (how == 0) {
            len2 = 0.0f + v[0] * v[0];
            len2 = len2 + v[1] * v[1];
            len2 = len2 + v[2] * v[2];
            d = sqrtf(len2);
        } else if (how == 1) {
            const float p0 = v[0] * v[0], p1 = v[1] * v[1], p2 = v[2] * v[2];
            double acc = 0.0 + (double)p0;
            acc = acc + (double)p1;
            acc = acc + (double)p2;
            len2 = (float)acc;
            d = sqrtf(len2);
        } else {
            h = splitmix64(h);
            uint32_t man = (uint32_t)h & 0x7fffffu;
            const uint32_t ex = 127u - 50u + (uint32_t)((h >> 23) % 84u);     // 2^-50 .. 2^33: a little beyond both ends
            if ((ctr & 3u) == 0u) man = ((h >> 40) & 1u) ? 0x7fffffu : 0u;
            d = __uint_as_float((ex << 23) | man);
            len2 = d * d;
        }
        const float wx = v[0] / d, wy = v[1] / d, wz = v[2] / d;
        {   // the scalar form with its own validity flags (grid traversal's shading step)
            float ux, uy, uz, amin = 1.0f;
            uint32_t hiflag = 0;
            div3_shared(v[0], v[1], v[2], d, len2, ux, uy, uz, amin, hiflag);
            if (!(amin < 0x1p-96f || (int32_t)hiflag < 0)) {
                ++accepted;
                if (__float_as_uint(ux) != __float_as_uint(wx) || __float_as_uint(uy) != __float_as_uint(wy) ||
                    __float_as_uint(uz) != __float_as_uint(wz)) { ++bad; if (first == ~0ull) first = ctr; }
            }
        }
        if (div3_operands_ok(len2, v[0], v[1], v[2])) {   // the packed form of the 8-sphere bounce block
            ++accepted;
            f2 uxy;
            float uz;
            div3_packed(f2{v[0], v[1]}, v[2], d, uxy, uz);
            if (__float_as_uint(uxy.x) != __float_as_uint(wx) || __float_as_uint(uxy.y) != __float_as_uint(wy) ||
                __float_as_uint(uz) != __float_as_uint(wz)) { ++bad; if (first == ~0ull) first = ctr; }
        }
    }
    if (bad) { atomicAdd(&result[0], bad); atomicMin(&result[1], first); }
    atomicAdd(&result[2], accepted);
#endif
}


} // namespace
