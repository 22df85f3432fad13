// pt_leaf.h -- numpy's pairwise summation of the S samples of a sub-pixel (np.mean at scripts/data_visualization.py:41-45 sums a
// float32 row with pairwise_sum: blocks of <= 128 elements are summed with 8 interleaved accumulators, larger ranges are halved
// recursively at a multiple of 8), flattened into a PLAN of leaves the kernels walk left to right: leaf i has len(i) samples and is
// followed by ncomb(i) "add the two topmost partial sums".  Plain C++: the kernels include it through pt_trace.h, the launch path
// (render_kernels.hip) and the CPU sanitizer driver (tests/sanitize/host_driver.cpp) build plans with make_leaf_plan().
#pragma once
#include <stdint.h>
#include <string.h>

#include <utility>
#include <vector>

#include "pt_core.h"   // APT_HD

namespace apt {

constexpr int kMaxLeaves = 64;   // pairwise-sum leaves of a plan
// Every sample count up to this one has a plan of <= kMaxLeaves leaves.  Above it the halving at multiples of 8 leaves some counts with
// 65 leaves (7689 ... 8191; 8192 itself is 64 leaves of 128): those are refused with APT_ERR_ARG.  (Found by the CPU sanitizer driver's
// own checks in round 4 -- earlier rounds' message claimed "max 8192" for all of them.)
constexpr uint32_t kMaxPlanSamples = 7688;

struct LeafProg { // numpy pairwise_sum recursion flattened (see build_leaves)
    uint32_t nleaves;
    uint32_t maxleaf;          // longest leaf (sizes the refill colour queue)
    uint32_t leaf[kMaxLeaves]; // len | ncomb << 16 (dwords: wave-uniform s_load from the kernarg segment)
    APT_HD uint32_t len(uint32_t i) const { return leaf[i] & 0xffffu; }
    APT_HD uint32_t ncomb(uint32_t i) const { return leaf[i] >> 16; }
};

inline void build_leaves(uint32_t n, std::vector<std::pair<uint32_t, uint32_t>> &out) {
    if (n <= 128) { out.push_back({n, 0u}); return; }
    uint32_t n2 = n / 2;
    n2 -= n2 % 8;
    build_leaves(n2, out);
    build_leaves(n - n2, out);
    out.back().second += 1;
}

// -> false when `samples` needs more than kMaxLeaves leaves (never for samples <= kMaxPlanSamples)
inline bool make_leaf_plan(uint32_t samples, LeafProg &lp) {
    std::vector<std::pair<uint32_t, uint32_t>> v;
    build_leaves(samples, v);
    if (v.size() > (size_t)kMaxLeaves) return false;
    memset(&lp, 0, sizeof lp);
    lp.nleaves = (uint32_t)v.size();
    for (size_t i = 0; i < v.size(); ++i) {
        lp.leaf[i] = v[i].first | (v[i].second << 16);
        lp.maxleaf = v[i].first > lp.maxleaf ? v[i].first : lp.maxleaf;
    }
    return true;
}

} // namespace apt
