// host_driver.cpp -- CPU sanitizer driver for the HOST side of librender_mi355x.so (tests/test_host_sanitizers.py builds it with
// -fsanitize=address,undefined together with csrc/host_helpers.cpp and oracle/pt_oracle.c; no GPU, no HIP).  It drives the code
// that indexes a lot of memory on the host: both forms of the uniform-grid tables (1e5 spheres, clustered, NaN / inf members, a
// flat slab, a single sphere), MT19937 checkpoint windows that chain, gen_rays_host, the pairwise-sum leaf plans for every sample
// count 1 ... 8192 (+ the refusal above), and a small frame of the oracle's C restatement.  Every result is also CHECKED (a
// sanitizer only sees the accesses a run makes): grid tables against their own invariants, chained windows against one pass.
// Prints "ok <checks>" and exits 0; a failed check prints what failed and exits 1; a sanitizer report aborts.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <limits>
#include <vector>

#include "../../include/render_mi355x.h"
#include "../../ascendpathtracing_amd/csrc/pt_core.h"
#include "../../ascendpathtracing_amd/csrc/pt_leaf.h"

extern "C" {   // oracle/pt_oracle.c (test infrastructure: run here under the sanitizers, as a checker of itself)
typedef apt_render_params oracle_params;   // same layout (oracle/oracle.py Params)
int oracle_gen_spheres(float *out);
int oracle_render_frame(const oracle_params *P, const float *sph, uint64_t pixel_begin, uint64_t pixel_count, float *fb, uint8_t *u8,
                        double *pre, int threads, uint64_t *traced_out);
int oracle_gen_rays(uint32_t w, uint32_t h, uint32_t s, uint32_t seed, float *rays);
}

static long g_checks = 0;
#define CHECK(cond, ...) do { ++g_checks; if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s -- ", __FILE__, __LINE__, #cond); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); exit(1); } } while (0)

// ---- grid tables ---------------------------------------------------------------------------------------------------------
static void check_grid(const char *name, std::vector<float> &scene, uint32_t ns) {
    size_t bytes = 0;
    CHECK(apt_build_grid_host(scene.data(), ns, nullptr, &bytes) == APT_OK && bytes >= sizeof(apt::GridHeader), "%s: size query", name);
    std::vector<uint32_t> w(bytes / 4 + 1, 0xdeadbeefu);      // one guard word behind the buffer
    size_t bytes2 = 0;
    CHECK(apt_build_grid_host(scene.data(), ns, w.data(), &bytes2) == APT_OK && bytes2 == bytes, "%s: build", name);
    CHECK(w[bytes / 4] == 0xdeadbeefu, "%s: wrote past the size it reported", name);
    apt::GridHeader h;
    memcpy(&h, w.data(), sizeof h);
    CHECK(h.magic == apt::kGridMagic && h.num_spheres == ns, "%s: header", name);
    CHECK(h.ncells == h.n[0] * h.n[1] * h.n[2] && h.ncells >= 1, "%s: cell count", name);
    const uint32_t *cells = w.data() + h.off_cells, *items = w.data() + h.off_items, *large = w.data() + h.off_large;
    CHECK(cells[0] == 0 && cells[h.ncells] == h.nitems, "%s: cell_start ends", name);
    std::vector<uint8_t> seen(ns, 0);
    for (uint32_t i = 0; i < h.nlarge; ++i) { CHECK(large[i] < ns, "%s: large id", name); seen[large[i]] |= 1; }
    for (uint32_t c = 0; c < h.ncells; ++c) {
        CHECK(cells[c] <= cells[c + 1], "%s: cell_start monotone at %u", name, c);
        for (uint32_t i = cells[c]; i < cells[c + 1]; ++i) {
            CHECK(items[i] < ns, "%s: item id", name);
            CHECK(i == cells[c] || items[i - 1] < items[i], "%s: ids ascend inside cell %u", name, c);
            seen[items[i]] |= 2;
        }
    }
    // every sphere is in the always-tested list or in at least one cell (a small sphere's own cell always "touches" it), never in both
    for (uint32_t k = 0; k < ns; ++k) CHECK(seen[k] == 1 || seen[k] == 2, "%s: sphere %u is listed %s", name, k, seen[k] ? "twice" : "nowhere");
    // item_geom mirrors geom[items[i]]; sphere8 mirrors the table
    const float *geom = (const float *)(w.data() + h.off_geom), *ig = (const float *)(w.data() + h.off_item_geom);
    for (uint32_t i = 0; i < h.nitems; ++i) CHECK(memcmp(ig + 4 * (size_t)i, geom + 4 * (size_t)items[i], 16) == 0, "%s: item_geom[%u]", name, i);
    if (h.off_cellslot) {
        const uint32_t *cellslot = w.data() + h.off_cellslot, *ids = w.data() + h.off_slot_ids;
        const float *sg = (const float *)(w.data() + h.off_slots);
        CHECK((size_t)h.off_sphere8 + 8 * (size_t)ns == bytes / 4, "%s: table end", name);
        // cellslot is indexed by bordered cell coordinates (round 4): the layer around the grid holds the "outside" mark
        const uint32_t sx = h.n[0] + 2, sy = h.n[1] + 2, sz = h.n[2] + 2;
        CHECK((size_t)h.off_cellslot + apt::grid_bordered_cells(h.n) <= h.off_slots, "%s: bordered cellslot table overlaps the slots", name);
        for (uint32_t t = 0; t < sx * sy * sz; ++t) {
            const uint32_t xb = t % sx, yb = (t / sx) % sy, zb = t / (sx * sy);
            if (xb == 0 || yb == 0 || zb == 0 || xb == sx - 1 || yb == sy - 1 || zb == sz - 1) CHECK(cellslot[t] == apt::kGridCellOutside, "%s: border entry %u", name, t);
        }
        for (uint32_t c = 0; c < h.ncells; ++c) {
            const uint32_t b = cells[c], e = cells[c + 1], n = (e - b + 1) >> 1;
            const uint32_t x = c % h.n[0], y = (c / h.n[0]) % h.n[1], z = c / (h.n[0] * h.n[1]);
            const uint32_t entry = cellslot[((z + 1) * sy + (y + 1)) * sx + (x + 1)];
            const uint32_t slot0 = entry >> apt::kGridSlotShift, cnt = entry & apt::kGridSlotCountMax;
            CHECK(slot0 == apt::grid_slot_begin(h, b, c) && cnt == (n < apt::kGridCellOutside ? n : apt::kGridSlotCountMax), "%s: cellslot of cell %u", name, c);
            CHECK((uint64_t)slot0 + n <= h.nslots, "%s: slots of cell %u beyond the table", name, c);
            for (uint32_t i = b; i < e; ++i) {                 // candidate i - b sits in slot slot0 + (i-b)/2, half (i-b)&1
                const uint32_t s = slot0 + ((i - b) >> 1), half = (i - b) & 1u;
                CHECK(ids[2 * (size_t)s + half] == items[i], "%s: slot id", name);
                const float *g = sg + 8 * (size_t)s, *src = geom + 4 * (size_t)items[i];
                CHECK(memcmp(&g[half], &src[0], 4) == 0 && memcmp(&g[2 + half], &src[1], 4) == 0 && memcmp(&g[4 + half], &src[2], 4) == 0 &&
                      memcmp(&g[6 + half], &src[3], 4) == 0, "%s: slot geometry", name);
            }
            if ((e - b) & 1u) CHECK(ids[2 * (size_t)(slot0 + n - 1) + 1] == apt::kGridNoSphere, "%s: pad id of cell %u", name, c);
        }
        for (uint32_t j = 0; j < h.nlarge; ++j) CHECK(ids[j] == large[j], "%s: always-tested slot ids", name);
    }
}

static std::vector<float> gen_scene(uint32_t ns, uint64_t seed) {
    size_t n = 0;
    CHECK(apt_gen_scene_host(ns, seed, nullptr, &n) == APT_OK && n >= (size_t)ns * 10 && n % 128 == 0, "scene size");
    std::vector<float> s(n);
    CHECK(apt_gen_scene_host(ns, seed, s.data(), &n) == APT_OK, "scene");
    return s;
}

static void grids() {
    { auto s = gen_scene(100000, 4); check_grid("1e5 spheres", s, 100000); }
    { auto s = gen_scene(10000, 1); check_grid("1e4 spheres", s, 10000); }
    {   // clustered: 2000 spheres within 0.02 of one spot + exact copies (lists far beyond the 63 slots a cell entry can count)
        const uint32_t ns = 3000;
        auto s = gen_scene(ns, 9);
        uint64_t st = 12345;
        for (uint32_t k = 100; k < 2100; ++k) {
            for (int a = 1; a <= 3; ++a) s[(size_t)a * ns + k] = (a == 1 ? 50.0f : a == 2 ? 42.0f : 110.0f) + (float)((double)(apt::xorshift64s(st) >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 0.04f;
            s[k] = 1.0f + (float)(k % 7) * 0.1f;
        }
        for (uint32_t k = 2100; k < 2200; ++k) for (int a = 0; a <= 3; ++a) s[(size_t)a * ns + k] = s[(size_t)a * ns + 100];
        check_grid("clustered", s, ns);
    }
    {   // non-finite members: NaN / inf centres and radii must land in the always-tested list
        const uint32_t ns = 500;
        auto s = gen_scene(ns, 2);
        const float nan = std::numeric_limits<float>::quiet_NaN(), inf = std::numeric_limits<float>::infinity();
        s[1 * ns + 10] = nan; s[2 * ns + 11] = inf; s[3 * ns + 12] = -inf; s[0 * ns + 13] = nan; s[0 * ns + 14] = inf; s[0 * ns + 15] = -1.0f;
        s[0 * ns + 16] = 0.0f;
        check_grid("non-finite", s, ns);
    }
    {   // flat slab: every small sphere in one plane (one cell layer), and a far-away pair (huge margin)
        const uint32_t ns = 400;
        auto s = gen_scene(ns, 3);
        for (uint32_t k = 6; k < ns - 1; ++k) s[2 * ns + k] = 40.0f;
        check_grid("flat slab", s, ns);
        s[1 * ns + 20] = 1.0e6f; s[1 * ns + 21] = -1.0e6f;
        check_grid("far away pair", s, ns);
    }
    {   // the smallest scenes: 8 (no small spheres at all) and 9
        auto s8 = gen_scene(8, 1); check_grid("ns 8", s8, 8);
        auto s9 = gen_scene(9, 1); check_grid("ns 9", s9, 9);
    }
    size_t b = 0;
    CHECK(apt_build_grid_host(nullptr, 10, nullptr, &b) == APT_ERR_ARG, "null scene refused");
}

// ---- MT19937 checkpoint windows ---------------------------------------------------------------------------------------------
static void mt_windows() {
    const uint32_t seed = 7, stride = 5;
    const uint64_t nblocks = 203;                                            // not a multiple of the stride
    const uint64_t ncp = (nblocks + stride - 1) / stride;
    std::vector<uint32_t> one(ncp * 624), st(624);
    CHECK(apt_mt19937_checkpoints_host(seed, nblocks, stride, one.data()) == APT_OK, "one pass");
    // the same table from three chained windows whose lengths are multiples of the stride, the second started from a stored state
    std::vector<uint32_t> chained(ncp * 624, 0), state(624), state2(624);
    const uint64_t w0 = 10 * stride, w1 = 20 * stride, w2 = nblocks - w0 - w1;
    CHECK(apt_mt19937_checkpoints_window(nullptr, seed, 0, w0, stride, chained.data(), state.data()) == APT_OK, "window 0");
    CHECK(apt_mt19937_checkpoints_window(state.data(), seed, w0, w1, stride, chained.data() + (w0 / stride) * 624, state2.data()) == APT_OK, "window 1");
    CHECK(apt_mt19937_checkpoints_window(state2.data(), seed, w0 + w1, w2, stride, chained.data() + ((w0 + w1) / stride) * 624, nullptr) == APT_OK, "window 2");
    CHECK(one == chained, "chained windows differ from one pass");
    // a far window twisted there from the seed equals the chained state
    std::vector<uint32_t> far(624);
    CHECK(apt_mt19937_checkpoints_window(nullptr, seed, w0 + w1, 1, 1, far.data(), nullptr) == APT_OK && far == state2, "far window");
    CHECK(apt_mt19937_checkpoints_window(nullptr, seed, 0, 0, 1, far.data(), nullptr) == APT_ERR_ARG, "empty window refused");
    CHECK(apt_mt19937_checkpoints_window(nullptr, seed, 0, 1, 0, far.data(), nullptr) == APT_ERR_ARG, "zero stride refused");
}

// ---- gen_rays_host against the oracle's own generator ---------------------------------------------------------------------------
static void rays() {
    for (uint32_t s : {1u, 2u, 5u}) {
        const uint32_t w = 7, h = 5;
        const size_t n = (size_t)w * h * 4 * s;
        std::vector<float> a(6 * n + 1, -7.0f), b(6 * n);
        CHECK(apt_gen_rays_host(w, h, s, 0, a.data()) == APT_OK && a[6 * n] == -7.0f, "gen_rays_host");
        CHECK(oracle_gen_rays(w, h, s, 0, b.data()) == 0, "oracle_gen_rays");
        CHECK(memcmp(a.data(), b.data(), 6 * n * 4) == 0, "host rays differ from the oracle's (S = %u)", s);
    }
    CHECK(apt_gen_rays_host(0, 1, 1, 0, nullptr) == APT_ERR_ARG, "bad arguments refused");
}

// ---- pairwise-sum leaf plans ----------------------------------------------------------------------------------------------------
static float pairwise_ref(const float *a, uint32_t n) {   // numpy's pairwise_sum (the recursion the plan flattens)
    if (n < 8) { float r = 0.0f; for (uint32_t i = 0; i < n; ++i) r = r + a[i]; return r; }
    if (n <= 128) {
        float r[8];
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        uint32_t i;
        for (i = 8; i < n - n % 8; i += 8) for (int k = 0; k < 8; ++k) r[k] = r[k] + a[i + k];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res = res + a[i];
        return res;
    }
    uint32_t n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_ref(a, n2) + pairwise_ref(a + n2, n - n2);
}
static void leaf_plans() {
    std::vector<float> a(8192);
    uint64_t st = 99;
    for (auto &x : a) x = (float)((double)(apt::xorshift64s(st) >> 11) * (1.0 / 9007199254740992.0));
    uint32_t refused = 0;
    for (uint32_t s = 1; s <= 8192; ++s) {
        apt::LeafProg lp;
        std::vector<std::pair<uint32_t, uint32_t>> v;
        apt::build_leaves(s, v);
        const bool fits = v.size() <= (size_t)apt::kMaxLeaves;
        CHECK(apt::make_leaf_plan(s, lp) == fits, "plan for S = %u", s);
        // every count up to kMaxPlanSamples fits; above it only some do (8192 = 64 leaves of 128 does, 7689 ... 8191 need 65 leaves)
        CHECK(fits || s > apt::kMaxPlanSamples, "S = %u refused below the documented limit", s);
        if (!fits) { ++refused; continue; }
        CHECK(lp.nleaves >= 1 && lp.nleaves <= (uint32_t)apt::kMaxLeaves && lp.maxleaf <= 128, "plan shape, S = %u", s);
        uint32_t total = 0;
        float stack[16];
        int sp = 0;
        for (uint32_t i = 0; i < lp.nleaves; ++i) {                                   // walk the plan as the kernels do
            CHECK(lp.len(i) >= 1 && lp.len(i) <= lp.maxleaf && (lp.nleaves == 1 || lp.len(i) >= 8), "leaf length, S = %u", s);
            CHECK(sp < 15, "stack depth, S = %u", s);
            stack[sp++] = pairwise_ref(a.data() + total, lp.len(i));
            total += lp.len(i);
            for (uint32_t m = 0; m < lp.ncomb(i); ++m) { CHECK(sp >= 2, "stack underflow, S = %u", s); --sp; stack[sp - 1] = stack[sp - 1] + stack[sp]; }
        }
        CHECK(total == s && sp == 1, "plan covers S = %u", s);
        if (s % 37 == 0 || s > 8100) {
            const float want = pairwise_ref(a.data(), s);
            CHECK(memcmp(&want, &stack[0], 4) == 0, "plan sum differs from numpy's recursion, S = %u", s);
        }
    }
    apt::LeafProg lp;
    CHECK(refused == 441 && apt::make_leaf_plan(8192, lp) && lp.nleaves == 64, "which counts above the limit are refused");
    CHECK(!apt::make_leaf_plan(8192 * 2 + 1, lp), "oversized sample count refused");
}

// ---- the oracle's C restatement on a small frame -------------------------------------------------------------------------
static void oracle_frame() {
    float sph[128];
    CHECK(oracle_gen_spheres(sph) == 0, "oracle_gen_spheres");
    float ours[128];
    CHECK(apt_gen_spheres_host(ours) == APT_OK && memcmp(sph, ours, sizeof sph) == 0, "scene tables agree");
    for (uint32_t flags : {0u, (uint32_t)APT_FLAG_RETIRE, (uint32_t)(APT_FLAG_RETIRE | APT_FLAG_RR)}) {
        apt_render_params p;
        apt_default_params(&p);
        p.width = 6; p.height = 5; p.samples = 20; p.depth = 9; p.flags = flags; p.seed = 3;
        const uint64_t npix = 30;
        std::vector<float> fb(3 * npix + 1, -3.0f);
        std::vector<uint8_t> u8(3 * npix + 1, 77);
        uint64_t traced = 0;
        CHECK(oracle_render_frame(&p, sph, 0, npix, fb.data(), u8.data(), nullptr, 2, &traced) == 0, "oracle_render_frame");
        CHECK(fb[3 * npix] == -3.0f && u8[3 * npix] == 77, "oracle wrote past its frame");
        CHECK(traced > 0 && traced <= npix * 4 * 20 * 9, "traced count");
        for (uint64_t i = 0; i < 3 * npix; ++i) CHECK(fb[i] >= 0.0f && fb[i] <= 1.0f, "pixel value out of [0,1]");
    }
}

int main() {
    grids();
    mt_windows();
    rays();
    leaf_plans();
    oracle_frame();
    printf("ok %ld\n", g_checks);
    return 0;
}
