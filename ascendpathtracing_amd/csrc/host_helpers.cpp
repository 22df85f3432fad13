// host_helpers.cpp -- host-side entry points of librender_mi355x.so that need no GPU:
//   apt_gen_rays_host     scripts/gen_data.py:21-75   gen_rays (MT19937 legacy stream, float64 camera)
//   apt_gen_spheres_host  scripts/gen_data.py:92-132  gen_spheres
//   apt_gen_scene_host    build-defined large scene (BASELINE config 4; no reference counterpart)
//   apt_write_ppm         scripts/data_visualization.py:11-17 write_ppm
// Compiled with -ffp-contract=off like the kernels (shared arithmetic: pt_core.h).
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include <string>

#include "../../include/render_mi355x.h"
#include "apt_host.h"
#include "pt_core.h"

// ---- error record and contexts (apt_host.h) ---------------------------------------------------------
namespace {
thread_local std::string t_err;
thread_local int t_status = APT_OK;
} // namespace

namespace apt {
void clear_error() { t_err.clear(); t_status = APT_OK; }
int set_error(int code, const char *fmt, const char *detail) {
    char buf[256];
    snprintf(buf, sizeof buf, fmt, detail);
    t_err = buf;
    t_status = code;
    return code;
}
apt_context &default_context() {
    static apt_context ctx;
    return ctx;
}
} // namespace apt

// The APT_* measurement knobs of the process environment are read HERE, once per context, and nowhere else: no launch path calls
// getenv() (it races with setenv() in another thread, and a per-context API must not depend on process-global state).
namespace {
double env_number(const char *name) {
    const char *e = getenv(name);
    return e ? atof(e) : 0.0;
}
} // namespace

apt_context::apt_context() {
    apt_default_params(&v_.params);
    v_.trace_counter = nullptr;
    v_.refill_lanes = apt::kDefaultRefillLanes;
    struct { const char *env, *key; } knobs[] = {{"APT_QUEUE_PPW", "queue_ppw"}, {"APT_QUEUE_NBUF", "queue_nbuf"}, {"APT_QUEUE_LDS_PAD", "queue_lds_pad"},
                                                 {"APT_GRID_SPHERES_PER_CELL", "grid_spheres_per_cell"}};
    for (const auto &k : knobs) {
        const double v = env_number(k.env);
        if (v != 0.0) (void)set_debug(k.key, v);     // out-of-range values are ignored, as before
    }
    if (const char *e = getenv("APT_GRID_WALK")) if (e[0] == 'i') v_.debug.grid_walk = 1;   // "items"
    apt::clear_error();
}
apt_context::Values apt_context::snapshot() { std::lock_guard<std::mutex> g(m_); return v_; }
void apt_context::set_params(const apt_render_params &p) { std::lock_guard<std::mutex> g(m_); v_.params = p; }
void apt_context::set_trace_counter(unsigned long long *c) { std::lock_guard<std::mutex> g(m_); v_.trace_counter = c; }
void apt_context::set_refill_lanes(uint32_t lanes) { std::lock_guard<std::mutex> g(m_); v_.refill_lanes = lanes; }
int apt_context::set_debug(const char *key, double value) {
    if (!key) return apt::set_error(APT_ERR_ARG, "apt_context_set_debug: key is null%s");
    const std::string k(key);
    const bool whole = value == std::floor(value);
    std::lock_guard<std::mutex> g(m_);
    apt::Debug &d = v_.debug;
    if (k == "queue_ppw") { if (!(whole && value >= 0 && value <= 4096)) goto range; d.queue_ppw = (uint32_t)value; }
    else if (k == "queue_nbuf") { if (!(whole && (value == 0 || (value >= 2 && value <= 16)))) goto range; d.queue_nbuf = (uint32_t)value; }
    else if (k == "queue_lds_pad") { if (!(whole && value >= 0 && value <= 32768)) goto range; d.queue_lds_pad = (uint32_t)value; }
    else if (k == "grid_walk") { if (!(value == 0 || value == 1)) goto range; d.grid_walk = (uint32_t)value; }
    else if (k == "grid_spheres_per_cell") { if (!(value == 0 || (value > 0.01 && value < 1e6))) goto range; d.grid_spheres_per_cell = value; }
    else return apt::set_error(APT_ERR_ARG, "apt_context_set_debug: unknown key '%s'", key);
    return APT_OK;
range:
    return apt::set_error(APT_ERR_ARG, "apt_context_set_debug: value out of range for '%s'", key);
}
int apt_context::get_debug(const char *key, double *value) {
    if (!key) return apt::set_error(APT_ERR_ARG, "apt_context_get_debug: key is null%s");
    const std::string k(key);
    std::lock_guard<std::mutex> g(m_);
    const apt::Debug &d = v_.debug;
    if (k == "queue_ppw") *value = d.queue_ppw;
    else if (k == "queue_nbuf") *value = d.queue_nbuf;
    else if (k == "queue_lds_pad") *value = d.queue_lds_pad;
    else if (k == "grid_walk") *value = d.grid_walk;
    else if (k == "grid_spheres_per_cell") *value = d.grid_spheres_per_cell;
    else return apt::set_error(APT_ERR_ARG, "apt_context_get_debug: unknown key '%s'", key);
    return APT_OK;
}
uint32_t *apt_context::status_lookup(int dev) {
    if (dev < 0 || dev >= apt::kMaxStatusDevices) return nullptr;
    std::lock_guard<std::mutex> g(m_);
    return status_[dev];
}
uint32_t *apt_context::status_adopt(int dev, uint32_t *fresh, uint32_t **spare) {
    *spare = nullptr;
    if (dev < 0 || dev >= apt::kMaxStatusDevices) { *spare = fresh; return nullptr; }
    std::lock_guard<std::mutex> g(m_);
    if (status_[dev]) *spare = fresh;
    else status_[dev] = fresh;
    return status_[dev];
}
void apt_context::status_release(uint32_t *out[apt::kMaxStatusDevices]) {
    std::lock_guard<std::mutex> g(m_);
    for (int i = 0; i < apt::kMaxStatusDevices; ++i) { out[i] = status_[i]; status_[i] = nullptr; }
}

namespace {
using apt::set_error;

// np.random.seed(s); np.random.rand(): MT19937 (init_genrand) and the 53-bit
// random_sample construction ((a >> 5) * 2^26 + (b >> 6)) / 2^53.
class Mt19937 {
  public:
    explicit Mt19937(uint32_t seed) {
        mt_[0] = seed;
        for (int i = 1; i < kN; ++i) mt_[i] = 1812433253u * (mt_[i - 1] ^ (mt_[i - 1] >> 30)) + (uint32_t)i;
        pos_ = kN;
    }
    uint32_t next32() {
        if (pos_ >= kN) refill();
        uint32_t y = mt_[pos_++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    double next_double() {
        const uint32_t a = next32() >> 5, b = next32() >> 6;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }

  private:
    static constexpr int kN = 624, kM = 397;
    void refill() {
        for (int i = 0; i < kN; ++i) {
            const uint32_t y = (mt_[i] & 0x80000000u) | (mt_[(i + 1) % kN] & 0x7fffffffu);
            mt_[i] = mt_[(i + kM) % kN] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        pos_ = 0;
    }
    uint32_t mt_[kN];
    int pos_;
};

// The reference scene table, (r, x, y, z, em*3, col*3) per sphere: gen_data.py:94-102.
const double kSpheres[8][10] = {
    {1e5, 1e5 + 1, 40.8, 81.6, 0, 0, 0, 0.435, 0.376, 0.667},     // Left
    {1e5, -1e5 + 99, 40.8, 81.6, 0, 0, 0, 0.667, 0.129, 0.086},   // Right
    {1e5, 50, 40.8, 1e5, 0, 0, 0, 0.270, 0.725, 0.486},           // Back
    {1e5, 50, 40.8, -1e5 + 170, 0, 0, 0, 0, 0, 0},                // Front (black)
    {1e5, 50, 1e5, 81.6, 0, 0, 0, 0.5, 0.5, 0.5},                 // Bottom
    {1e5, 50, -1e5 + 81.6, 81.6, 0, 0, 0, 0.141, 0.408, 0.635},   // Top
    {16.5, 27, 16.5, 47, 0, 0, 0, 0.999, 0.999, 0.999},           // Mirror
    {600, 50, 681.6 - 0.27, 81.6, 12, 12, 12, 0, 0, 0}};          // Light

void sphere_record(int k, float rec[10]) {
    for (int m = 0; m < 10; ++m) {
        double v = kSpheres[k][m];
        if (m == 0) v = v * v; // gen_data.py:109: r -> r^2 in float64, float32 only at tofile (:127)
        rec[m] = (float)v;
    }
}

size_t padded_floats(size_t n) { return (n + 127) / 128 * 128; } // gen_data.py:120-127: 512-byte multiple

} // namespace

extern "C" {

const char *apt_last_error(void) { return t_err.c_str(); }
int apt_last_status(void) { return t_status; }

void apt_default_params(apt_render_params *p) {
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->struct_size = sizeof *p;
    p->width = 16; p->height = 16; p->samples = 1; // common.h:4-6
    p->depth = 5;                                   // render.cpp:141
    p->num_spheres = 8; p->light_index = 7;         // common.h:10, rt_helper.h:776
    p->eps = 1e-4f; p->gain = 12.0f;                // common.h:9, render.cpp:194
    p->mode = APT_MODE_KERNEL;
}

int apt_gen_rays_host(uint32_t width, uint32_t height, uint32_t samples, uint32_t seed, float *rays) {
    apt::clear_error();
    if (!rays || !width || !height || !samples) return set_error(APT_ERR_ARG, "apt_gen_rays_host: rays must be non-null, width/height/samples non-zero%s");
    Mt19937 rng(seed); // np.random.seed(0): gen_data.py:438
    apt::Camera cam;
    apt::camera_init(cam, width, height);
    const uint64_t n = (uint64_t)width * height * 4u * samples;
    uint64_t p = 0;
    for (uint32_t i = 0; i < width; ++i)               // gen_data.py:32-36 loop nest
        for (uint32_t j = 0; j < height; ++j)
            for (uint32_t sy = 0; sy < 2; ++sy)
                for (uint32_t sx = 0; sx < 2; ++sx)
                    for (uint32_t k = 0; k < samples; ++k, ++p) {
                        const double u1 = rng.next_double(); // r1 before r2: :37,:39
                        const double u2 = rng.next_double();
                        const apt::Ray ray = apt::camera_ray(cam, width, height, i, j, sy, sx, u1, u2);
                        rays[p] = ray.ox; rays[n + p] = ray.oy; rays[2 * n + p] = ray.oz;   // SoA: :65-71
                        rays[3 * n + p] = ray.dx; rays[4 * n + p] = ray.dy; rays[5 * n + p] = ray.dz;
                    }
    return APT_OK;
}

// Checkpoints of the MT19937 stream for the device generator (apt_gen_rays_mt_device).  Output
// block b (624 words = the 4 words of 156 consecutive paths) is the tempering of the state after
// b+1 twists; checkpoint i is that raw state for block i*stride.  Sequential by nature; done once
// per (seed, length) and reusable for every shorter length.
int apt_mt19937_checkpoints_window(const uint32_t *state_in, uint32_t seed, uint64_t first_block, uint64_t num_blocks,
                                   uint32_t stride, uint32_t *states, uint32_t *state_out) {
    apt::clear_error();
    if (!states || stride == 0 || num_blocks == 0) return set_error(APT_ERR_ARG, "apt_mt19937_checkpoints_window: states must be non-null, stride and num_blocks non-zero%s");
    uint32_t mt[624];
    auto twist = [&]() { // genrand twist
        for (int i = 0; i < 624; ++i) {
            const uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
            mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
    };
    if (state_in) memcpy(mt, state_in, sizeof mt);
    else {
        mt[0] = seed;
        for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        for (uint64_t b = 0; b <= first_block; ++b) twist(); // block b is the tempering of the state after b+1 twists
    }
    for (uint64_t k = 0; k < num_blocks; ++k) {               // mt = raw state of block first_block + k
        if (k % stride == 0) memcpy(states + (k / stride) * 624, mt, sizeof mt);
        if (k + 1 < num_blocks || state_out) twist();
    }
    if (state_out) memcpy(state_out, mt, sizeof mt);
    return APT_OK;
}

int apt_mt19937_checkpoints_host(uint32_t seed, uint64_t num_blocks, uint32_t stride, uint32_t *states) {
    return apt_mt19937_checkpoints_window(nullptr, seed, 0, num_blocks, stride, states, nullptr);
}

int apt_gen_spheres_host(float *spheres128) {
    apt::clear_error();
    if (!spheres128) return set_error(APT_ERR_ARG, "apt_gen_spheres_host: spheres128 is null%s");
    memset(spheres128, 0, 128 * sizeof(float));
    for (int k = 0; k < 8; ++k) {
        float rec[10];
        sphere_record(k, rec);
        for (int m = 0; m < 10; ++m) spheres128[m * 8 + k] = rec[m]; // transpose to planes: :113
    }
    return APT_OK;
}

int apt_gen_scene_host(uint32_t num_spheres, uint64_t seed, float *spheres, size_t *out_floats) {
    apt::clear_error();
    if (num_spheres < 8) return set_error(APT_ERR_SCENE, "apt_gen_scene_host: needs num_spheres >= 8 (six walls, at least one sphere, the light)%s");
    const size_t total = padded_floats((size_t)num_spheres * 10);
    if (out_floats) *out_floats = total;
    if (!spheres) return APT_OK;
    memset(spheres, 0, total * sizeof(float));
    for (uint32_t k = 0; k < num_spheres; ++k) {
        float rec[10];
        if (k < 6) sphere_record((int)k, rec);                 // the six walls
        else if (k == num_spheres - 1) sphere_record(7, rec);  // the light keeps index Ns-1
        else {                                                 // small random spheres inside the room
            uint64_t s = apt::splitmix64(seed ^ apt::splitmix64(0x5CE7E000ull + k));
            if (s == 0) s = 0x9E3779B97F4A7C15ull;
            double u[7];
            for (int m = 0; m < 7; ++m) u[m] = (double)(apt::xorshift64s(s) >> 11) * (1.0 / 9007199254740992.0);
            const double r = 0.5 + 1.5 * u[0];
            rec[0] = (float)(r * r);
            rec[1] = (float)(1.0 + 98.0 * u[1]);
            rec[2] = (float)(81.6 * u[2]);
            rec[3] = (float)(170.0 * u[3]);
            rec[4] = rec[5] = rec[6] = 0.0f;
            rec[7] = (float)(0.1 + 0.899 * u[4]);
            rec[8] = (float)(0.1 + 0.899 * u[5]);
            rec[9] = (float)(0.1 + 0.899 * u[6]);
        }
        for (int m = 0; m < 10; ++m) spheres[(size_t)m * num_spheres + k] = rec[m];
    }
    return APT_OK;
}

// Uniform grid for scenes with many spheres (apt_render_params.accel).  Spheres whose radius exceeds 8x the median
// radius, or that are not finite (pt_core.h grid_is_large), are "large" (the walls and the light of the generated scenes, r >= 600):
// they go to an always-tested list; the others are binned by their bounding boxes, inflated by `margin`
// so that any ray the fp32 intersection formula can possibly accept passes through the interior of a
// cell that lists the sphere (the formula's absolute error on disc is ~1e-3 at these coordinates; the
// margin is 0.05 + 1e-4 * coordinate scale).  Cells are sized for kGridSpheresPerCell = 0.5 sphere centres each (apt_set_debug("grid_spheres_per_cell", v) overrides: tuning knob).
uint32_t apt_grid_flags(const void *grid_head_host, uint32_t num_spheres) {
    apt::clear_error();
    if (!grid_head_host) return 0u;
    apt::GridHeader h;
    memcpy(&h, grid_head_host, sizeof h);
    return (h.magic == apt::kGridMagic && h.num_spheres == num_spheres && h.off_cellslot != 0u) ? (uint32_t)APT_FLAG_GRID_SLOTS : 0u;
}

int apt_build_grid_host(const float *sph, uint32_t ns, void *grid, size_t *out_bytes) {
    apt::clear_error();
    if (!sph || ns == 0 || !out_bytes) return set_error(APT_ERR_ARG, "apt_build_grid_host: spheres/out_bytes must be non-null, num_spheres non-zero%s");
    const float *r2 = sph, *cx = sph + ns, *cy = sph + 2 * (size_t)ns, *cz = sph + 3 * (size_t)ns;
    std::vector<float> rad(ns);
    for (uint32_t k = 0; k < ns; ++k) rad[k] = apt::grid_radius(r2[k]);
    std::vector<float> sorted(rad);
    std::nth_element(sorted.begin(), sorted.begin() + ns / 2, sorted.end());
    const float median = sorted[ns / 2];
    std::vector<uint32_t> large, small;
    for (uint32_t k = 0; k < ns; ++k)
        (apt::grid_is_large(r2[k], cx[k], cy[k], cz[k], rad[k], median) ? large : small).push_back(k);
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f}, scale = 1.0f;
    for (uint32_t k : small) {
        const float c[3] = {cx[k], cy[k], cz[k]};
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], c[a] - rad[k]); hi[a] = std::max(hi[a], c[a] + rad[k]); scale = std::max(scale, std::fabs(c[a]) + rad[k]); }
    }
    // sphere centres per cell (boxes overlap ~4 cells each).  The nested walk (round 1, 64 spp): 74.4 / 70.7 / 70.3 / 70.8 / 72.2 ms at 2 / 1 / 0.7 /
    // 0.5 / 0.35; the sample-queue kernel's grid form (round 3): 66.8 / 60.1 / 58.8 / 58.4 / 59.0 / 60.5 / 63.6 ms at 2 / 1 / 0.7 / 0.5 / 0.35 / 0.25 /
    // 0.18 -- a cell step is cheaper there than a pair slot, so smaller cells pay a little longer.
    const double knob = apt::default_context().snapshot().debug.grid_spheres_per_cell;   // apt_set_debug("grid_spheres_per_cell", v): tuning knob
    const double per_cell = knob > 0.0 ? knob : apt::kGridSpheresPerCell;
    apt::GridHeader h;
    apt::grid_header_from_stats(ns, (uint32_t)small.size(), (uint32_t)large.size(), lo, hi, scale, per_cell, h);
    std::vector<uint32_t> count(h.ncells + 1, 0);
    for (uint32_t k : small) {
        uint32_t x0, x1, y0, y1, z0, z1;
        apt::grid_cell_range(h, cx[k], rad[k], 0, x0, x1); apt::grid_cell_range(h, cy[k], rad[k], 1, y0, y1); apt::grid_cell_range(h, cz[k], rad[k], 2, z0, z1);
        for (uint32_t z = z0; z <= z1; ++z) for (uint32_t y = y0; y <= y1; ++y) for (uint32_t x = x0; x <= x1; ++x)
            if (apt::grid_cell_touches(h, cx[k], cy[k], cz[k], rad[k], x, y, z)) ++count[(z * h.n[1] + y) * h.n[0] + x + 1];
    }
    for (uint32_t c = 0; c < h.ncells; ++c) count[c + 1] += count[c];
    const size_t words = apt::grid_header_offsets(h, count[h.ncells]);
    *out_bytes = words * 4;
    if (!grid) return APT_OK;
    uint32_t *w = (uint32_t *)grid;
    memset(w, 0, words * 4);
    memcpy(w, &h, sizeof h);
    for (size_t i = 0; i < large.size(); ++i) w[h.off_large + i] = large[i];
    memcpy(w + h.off_cells, count.data(), (h.ncells + 1) * 4);
    std::vector<uint32_t> cursor(count.begin(), count.end() - 1);
    for (uint32_t k : small) {                                                       // ascending sphere index inside a cell
        uint32_t x0, x1, y0, y1, z0, z1;
        apt::grid_cell_range(h, cx[k], rad[k], 0, x0, x1); apt::grid_cell_range(h, cy[k], rad[k], 1, y0, y1); apt::grid_cell_range(h, cz[k], rad[k], 2, z0, z1);
        for (uint32_t z = z0; z <= z1; ++z) for (uint32_t y = y0; y <= y1; ++y) for (uint32_t x = x0; x <= x1; ++x)
            if (apt::grid_cell_touches(h, cx[k], cy[k], cz[k], rad[k], x, y, z)) w[h.off_items + cursor[(z * h.n[1] + y) * h.n[0] + x]++] = k;
    }
    float *g = (float *)(w + h.off_geom);
    for (uint32_t k = 0; k < ns; ++k) { g[4 * k] = cx[k]; g[4 * k + 1] = cy[k]; g[4 * k + 2] = cz[k]; g[4 * k + 3] = r2[k]; }
    float *ig = (float *)(w + h.off_item_geom);
    for (uint32_t i = 0; i < h.nitems; ++i) memcpy(ig + 4 * (size_t)i, g + 4 * (size_t)w[h.off_items + i], 16);
    if (h.off_cellslot) {                                                                             // pair-slot tables of the flat walk
        for (uint32_t t = 0, nb = apt::grid_bordered_cells(h.n); t <= nb; ++t) apt::grid_fill_cell_slots(w, h, t);
        for (uint32_t k = 0; k < ns; ++k) apt::grid_fill_sphere8(w, h, sph, k);
    }
    return APT_OK;
}

// write_ppm: the reference's loop `for i in range(w): for j in range(h): data[j, i]` over the
// (w,h,3) array returned by decode_color (second index already y-flipped) emits file row i
// = image row y = h-1-i and file column j = x.  It only stays in range for w == h; the
// non-square case is defined here the evident way, h rows of w pixels.
int apt_write_ppm(const char *path, uint32_t width, uint32_t height, const uint8_t *fb_u8) {
    apt::clear_error();
    if (!path || !fb_u8 || !width || !height) return set_error(APT_ERR_ARG, "apt_write_ppm: path/fb_u8 must be non-null, width/height non-zero%s");
    FILE *f = fopen(path, "w");
    if (!f) return set_error(APT_ERR_IO, "apt_write_ppm: cannot open %s", path);
    fprintf(f, "P3\n%u %u\n255\n", width, height);
    for (uint32_t row = 0; row < height; ++row) {
        const uint32_t y = height - 1 - row;
        for (uint32_t x = 0; x < width; ++x) {
            const uint8_t *px = fb_u8 + ((uint64_t)x * height + y) * 3;
            fprintf(f, "%u %u %u ", px[0], px[1], px[2]);
        }
        fprintf(f, "\n");
    }
    return fclose(f) == 0 ? APT_OK : set_error(APT_ERR_IO, "apt_write_ppm: write to %s failed", path);
}

} // extern "C"
