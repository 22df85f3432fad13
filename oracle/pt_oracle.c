/*
 * pt_oracle.c -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or
 * call this file.  The product (ascendpathtracing_amd/, librender_mi355x.so) never does.
 *
 * What is restated (paths relative to the reference repository root):
 *   oracle_render_paths   src/render.cpp:104-207 (KernelRender::Compute) with
 *                         src/rt_helper.h:255-370 (SphereHitInfo), :372-451
 *                         (Transpose + ReduceMinInfo), :504-709 (GenerateNewRays),
 *                         :711-830 (AccumulateIntervalColor)            -> K-mode
 *                         scripts/gen_data.py:190-243 (sim_npu), :246-429 (test_soa)
 *                                                                        -> O-mode
 *   oracle_test_scene     scripts/gen_data.py:134-188 (test_scene)
 *   oracle_gen_rays       scripts/gen_data.py:21-75 (gen_rays) under np.random.seed(seed)
 *   oracle_gen_spheres    scripts/gen_data.py:92-132 (gen_spheres)
 *   oracle_decode_color   scripts/data_visualization.py:20-59 (decode_color)
 *   oracle_write_ppm      scripts/data_visualization.py:11-17 (write_ppm)
 *   oracle_render_frame   composition gen_rays (counter RNG) -> render -> decode_color,
 *                         the restatement of the product's fused device entry
 *
 * Parity pin: tests/test_oracle_golden.py checks every function above against
 * tests/golden/golden.npz + golden.json, which were produced by RUNNING the reference's
 * own NumPy oracle (tests/golden/make_golden.py).  O-mode is bitwise equal to
 * test_soa.bin on every golden case.  The vendor (CANN) arithmetic behind K-mode cannot
 * be executed here: K-mode parity with real Ascend hardware is unpinned;
 * K-mode is pinned to the oracle only where both orders agree (depth <= 2 bitwise).
 *
 * Build: -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).  Every fp32 operation
 * below is a separately rounded IEEE operation; fma() is used only where NumPy's BLAS
 * provably uses one (float64 ddot, measured in this container: DESIGN.md "numerics").
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define MODE_K 0
#define MODE_O 1
#define FLAG_RETIRE 1u
#define FLAG_RR 2u
#define FLAG_EMISSION 4u

typedef struct {
    uint32_t struct_size, width, height, samples, depth, num_spheres;
    int32_t light_index;
    float eps, gain;
    uint32_t mode, flags, rr_start;
    uint64_t path_begin, path_count, seed, accel; /* accel: ignored here (brute force is the reference) */
} oracle_params; /* same layout as apt_render_params (include/render_mi355x.h) */

/* ------------------------------------------------------------------------------------ */
static uint64_t splitmix64(uint64_t x);
/* Russian roulette (extension; spec in include/render_mi355x.h, APT_FLAG_RR) */
static void russian_roulette(float *rx, float *ry, float *rz, int alive, uint64_t key, uint32_t bounce) {
    if (!alive) return;
    float q = *rx;
    if (*ry > q) q = *ry;
    if (*rz > q) q = *rz;
    if (!(q > 0.0f)) return;
    float p = q < 0.05f ? 0.05f : q;
    p = p > 0.95f ? 0.95f : p;
    uint64_t h = splitmix64(key + 0x9E3779B97F4A7C15ull * (uint64_t)(bounce + 1u));
    float u = (float)(uint32_t)(h >> 40) * 0x1p-24f;
    if (u >= p) { *rx = 0.0f; *ry = 0.0f; *rz = 0.0f; }
    else { float inv = 1.0f / p; *rx = *rx * inv; *ry = *ry * inv; *rz = *rz * inv; }
}

/* one path, all bounces.  sph = [10][Ns] planes.                                        */
/* returns the number of segments actually traced (== depth unless FLAG_RETIRE).         */
static uint32_t trace_path(const oracle_params *P, const float *sph, uint64_t path, float ox, float oy, float oz,
                           float dx, float dy, float dz, float out[3]) {
    const uint32_t rr_start = (P->flags & FLAG_RR) ? (P->rr_start ? P->rr_start : 3u) : 0u;
    const uint64_t rr_key = rr_start ? splitmix64(P->seed ^ splitmix64(path)) : 0;
    const uint32_t Ns = P->num_spheres;
    const float *r2 = sph, *cx = sph + Ns, *cy = sph + 2 * (size_t)Ns, *cz = sph + 3 * (size_t)Ns;
    const float *colx = sph + 7 * (size_t)Ns, *coly = sph + 8 * (size_t)Ns, *colz = sph + 9 * (size_t)Ns;
    const float eps = P->eps;
    float retx = 1.0f, rety = 1.0f, retz = 1.0f; /* render.cpp:116-121 */
    int alive = 1;                                /* render.cpp:123-124 retMask = all ones */
    uint32_t traced = 0;

    for (uint32_t depth = 0; depth < P->depth; ++depth) { /* render.cpp:140-188 */
        if ((P->flags & FLAG_RETIRE) && (!alive || (retx == 0.0f && rety == 0.0f && retz == 0.0f)))
            break; /* result preserving: SURVEY.md Appendix A notes */
        ++traced;
        /* ---- ComputeHitInfo: rt_helper.h:453-502; gen_data.py:276-321 ---- */
        float tmin = 1e20f;
        int64_t idx = (P->mode == MODE_O) ? -1 : 0; /* all-miss: gen_data.py:311 vs rt_helper.h:183-201 */
        for (uint32_t k = 0; k < Ns; ++k) {
            /* SphereHitInfo rt_helper.h:263-268: -(o + (-c)) == c - o exactly */
            float ocx = cx[k] - ox, ocy = cy[k] - oy, ocz = cz[k] - oz;
            /* :273,:297 Duplicate(0) start omitted: exact except for the sign of a zero that
             * cannot reach t (b only enters through b*b and b -/+ q); sim_npu omits it too. */
            float b = ocx * dx;            /* :278-280 FakeMulAddDst: mul, then add */
            b = b + ocy * dy;
            b = b + ocz * dz;
            float c = ocx * ocx;           /* :301-303 */
            c = c + ocy * ocy;
            c = c + ocz * ocz;
            c = c - r2[k];                 /* :304 Adds(c, c, -r2) */
            float disc = b * b;            /* :314 */
            disc = disc - c;               /* :315 */
            float q = sqrtf(disc);         /* :325 NaN for disc < 0 */
            float t0 = b - q, t1 = b + q;  /* :330-331 */
            float t = (t0 > eps) ? t0 : t1;  /* :341 FakeSelect; NaN compares false */
            t = (t > eps) ? t : 1e20f;       /* :349,:363 FakeCompare + Select */
            /* Transpose+ReduceMinInfo rt_helper.h:372-451: row minimum, lowest index among
             * equals; gen_data.py:312-321 strict '<' in ascending order: the same rule. */
            if (t < tmin) { tmin = t; idx = (int64_t)k; }
        }
        /* K-mode all-miss: every t == 1e20 == min, every compare bit set, lowest bit -> 0. */
        const uint32_t g = (idx < 0) ? Ns - 1 : (uint32_t)idx; /* Python index -1 wraps */

        /* ---- GenerateNewRays: rt_helper.h:504-709; gen_data.py:336-349 ---- */
        float hx = ox + dx * tmin, hy = oy + dy * tmin, hz = oz + dz * tmin; /* :513-518 */
        float nx = hx - cx[g], ny = hy - cy[g], nz = hz - cz[g];             /* :635-637 */
        float L, dot;
        if (P->mode == MODE_O) {
            /* np.linalg.norm (gen_data.py:347) = sqrt(sdot(n,n)); OpenBLAS sdot accumulates
             * the float32-rounded products in float64 and rounds once. */
            double acc = 0.0;
            acc += (double)(nx * nx); acc += (double)(ny * ny); acc += (double)(nz * nz);
            L = sqrtf((float)acc);
        } else {
            float s = 0.0f + nx * nx; s = s + ny * ny; s = s + nz * nz;      /* :641 Duplicate(0), :647-649 */
            L = sqrtf(s);                                                    /* :658 */
        }
        float ux = nx / L, uy = ny / L, uz = nz / L;                         /* :664-666 */
        if (P->mode == MODE_O) {
            double acc = 0.0;                                                /* np.dot gen_data.py:349 */
            acc += (double)(dx * ux); acc += (double)(dy * uy); acc += (double)(dz * uz);
            dot = (float)acc;
        } else {
            dot = 0.0f + dx * ux; dot = dot + dy * uy; dot = dot + dz * uz;  /* :690 Duplicate(0), :694-696 */
        }
        float k2 = dot * 2.0f;                                               /* :697 */
        dx = dx - ux * k2; dy = dy - uy * k2; dz = dz - uz * k2;             /* :699-703 */
        ox = hx; oy = hy; oz = hz;                                           /* :706-708 */

        /* ---- AccumulateIntervalColor: rt_helper.h:711-830; gen_data.py:379-390 ---- */
        if (idx == (int64_t)P->light_index) alive = 0;                       /* :773-787 */
        if (alive) { retx = colx[g] * retx; rety = coly[g] * rety; retz = colz[g] * retz; } /* :799-810 */
        if (rr_start && depth + 1 >= rr_start) russian_roulette(&retx, &rety, &retz, alive, rr_key, depth);
    }
    if ((P->flags & FLAG_EMISSION) && P->light_index >= 0) { /* emission planes 4..6 instead of the literal 12 */
        const size_t l = (size_t)P->light_index;
        out[0] = retx * sph[4 * (size_t)Ns + l]; out[1] = rety * sph[5 * (size_t)Ns + l]; out[2] = retz * sph[6 * (size_t)Ns + l];
    } else {
        out[0] = retx * P->gain; out[1] = rety * P->gain; out[2] = retz * P->gain; /* render.cpp:194-196 */
    }
    return traced;
}

/* rays [6][N], colors [3][N]; renders paths [path_begin, path_begin+path_count). */
int oracle_render_paths(const oracle_params *P, const float *rays, const float *sph, float *colors,
                        int threads, uint64_t *traced_out) {
    if (!P || !rays || !sph || !colors || P->num_spheres == 0) return 1;
    const uint64_t N = (uint64_t)P->width * P->height * 4u * P->samples;
    const uint64_t b = P->path_begin, n = P->path_count ? P->path_count : N - b;
    if (b + n > N) return 1;
    uint64_t traced = 0;
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1) reduction(+ : traced)
#endif
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        const uint64_t p = b + (uint64_t)i;
        float out[3];
        traced += trace_path(P, sph, p, rays[p], rays[N + p], rays[2 * N + p], rays[3 * N + p], rays[4 * N + p],
                             rays[5 * N + p], out);
        colors[p] = out[0]; colors[N + p] = out[1]; colors[2 * N + p] = out[2];
    }
    if (traced_out) *traced_out = traced;
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* scripts/gen_data.py:134-188 test_scene: first hit only; emission for the light,       */
/* colour otherwise, 0 when nothing is hit.  out = [3][N].                               */
int oracle_test_scene(const oracle_params *P, const float *rays, const float *sph, float *out) {
    const uint32_t Ns = P->num_spheres;
    const uint64_t N = (uint64_t)P->width * P->height * 4u * P->samples;
    for (uint64_t p = 0; p < N; ++p) {
        float o[3] = {rays[p], rays[N + p], rays[2 * N + p]};
        float d[3] = {rays[3 * N + p], rays[4 * N + p], rays[5 * N + p]};
        float mind = 1e20f; /* gen_data.py:147 (python float, weakly typed against float32) */
        int id = -1;
        for (uint32_t k = 0; k < Ns; ++k) {
            float op[3] = {sph[Ns + k] - o[0], sph[2 * Ns + k] - o[1], sph[3 * Ns + k] - o[2]}; /* :151 */
            double acc = 0.0; /* np.dot float32: float64 accumulation of float32 products */
            acc += (double)(op[0] * d[0]); acc += (double)(op[1] * d[1]); acc += (double)(op[2] * d[2]);
            float b = (float)acc;                                   /* :153 */
            acc = 0.0;
            acc += (double)(op[0] * op[0]); acc += (double)(op[1] * op[1]); acc += (double)(op[2] * op[2]);
            float det = b * b - (float)acc;                         /* :154 */
            det = det + sph[k];
            if (det < 0) continue;                                  /* :155 */
            det = sqrtf(det);
            float t0 = b - det, t1 = b + det;
            if (t0 > P->eps && t0 < mind) { mind = t0; id = (int)k; }       /* :163-165 */
            else if (t1 > P->eps && t1 < mind) { mind = t1; id = (int)k; }  /* :166-168 */
        }
        for (int c = 0; c < 3; ++c) {
            float v = 0.0f;
            if (id >= 0) v = (id == P->light_index) ? sph[(4 + c) * Ns + id] : sph[(7 + c) * Ns + id]; /* :175-180 */
            out[(uint64_t)c * N + p] = v;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* MT19937 (legacy np.random.seed / np.random.rand stream)                               */
typedef struct { uint32_t mt[624]; int idx; } mt19937;
static void mt_seed(mt19937 *m, uint32_t s) {
    m->mt[0] = s;
    for (int i = 1; i < 624; ++i) m->mt[i] = 1812433253u * (m->mt[i - 1] ^ (m->mt[i - 1] >> 30)) + (uint32_t)i;
    m->idx = 624;
}
static uint32_t mt_next(mt19937 *m) {
    if (m->idx >= 624) {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = (m->mt[i] & 0x80000000u) | (m->mt[(i + 1) % 624] & 0x7fffffffu);
            m->mt[i] = m->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        m->idx = 0;
    }
    uint32_t y = m->mt[m->idx++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}
static double mt_double(mt19937 *m) { /* random_sample: 53-bit */
    uint32_t a = mt_next(m) >> 5, b = mt_next(m) >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

/* counter-based generator of the device ray-generate path: the two 52-bit uniforms of path p are
 * outputs 2p+1 and 2p+2 of the SplitMix64 stream seeded with splitmix64(seed) (random access by
 * state = splitmix64(seed) + 2p*phi).  Pure integer arithmetic.  xorshift64* serves gen_scene. */
static uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
static uint64_t xorshift64s(uint64_t *s) {
    uint64_t x = *s;
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    *s = x;
    return x * 0x2545F4914F6CDD1Dull;
}
static double unit_from_bits(uint64_t z) { /* 52 high bits as the mantissa of a double in [1,2), minus 1 */
    const uint64_t b = (z >> 12) | 0x3FF0000000000000ull;
    double d;
    memcpy(&d, &b, sizeof d);
    return d - 1.0;
}
static void path_uniforms(uint64_t seed, uint64_t path, double *u1, double *u2) {
    /* outputs 2p+1 and 2p+2 of the SplitMix64 stream whose state starts at splitmix64(seed) */
    const uint64_t state = splitmix64(seed) + path * 0x3C6EF372FE94F82Aull; /* 2*phi mod 2^64 */
    *u1 = unit_from_bits(splitmix64(state));
    *u2 = unit_from_bits(splitmix64(state + 0x9E3779B97F4A7C15ull));
}

/* test hook: the uniforms of paths [first, first+n) -> out[2n] (u1, u2 interleaved) */
void oracle_path_uniforms(uint64_t seed, uint64_t first, uint64_t n, double *out) {
    for (uint64_t i = 0; i < n; ++i) path_uniforms(seed, first + i, out + 2 * i, out + 2 * i + 1);
}

/* camera frame of gen_rays (gen_data.py:24-30), all float64 */
typedef struct { double pos[3], g[3], cx[3], cy[3]; } camera;
static double norm3(const double v[3]) { /* np.linalg.norm = sqrt(ddot); ddot is an FMA chain here */
    double acc = v[0] * v[0];
    acc = fma(v[1], v[1], acc);
    acc = fma(v[2], v[2], acc);
    return sqrt(acc);
}
static void camera_init(camera *c, uint32_t w, uint32_t h) {
    c->pos[0] = 50; c->pos[1] = 52; c->pos[2] = 295.6;       /* :24 */
    double dir[3] = {0, -0.042612, -1};
    double n = norm3(dir);                                    /* :25 */
    for (int i = 0; i < 3; ++i) c->g[i] = dir[i] / n;
    c->cx[0] = (double)w * 0.5135 / (double)h; c->cx[1] = 0; c->cx[2] = 0;   /* :28 */
    double cr[3];                                             /* np.cross(cx, g) :29 */
    cr[0] = c->cx[1] * c->g[2] - c->cx[2] * c->g[1];
    cr[1] = c->cx[2] * c->g[0] - c->cx[0] * c->g[2];
    cr[2] = c->cx[0] * c->g[1] - c->cx[1] * c->g[0];
    double cn = norm3(cr);
    for (int i = 0; i < 3; ++i) c->cy[i] = cr[i] / cn * 0.5135;
}
static double tent(double u) { /* gen_data.py:37-40 */
    double r = 2 * u;
    return (r < 1) ? sqrt(r) - 1 : 1 - sqrt(2 - r);
}
static void camera_ray(const camera *c, uint32_t w, uint32_t h, uint32_t i, uint32_t j, uint32_t sy, uint32_t sx,
                       double u1, double u2, float ray[6]) {
    double ddx = tent(u1), ddy = tent(u2);
    double a = (((double)sx + 0.5 + ddx) / 2 + (double)i) / (double)w - 0.5;   /* :41 */
    double b = (((double)sy + 0.5 + ddy) / 2 + (double)j) / (double)h - 0.5;   /* :42 */
    double d[3];
    for (int k = 0; k < 3; ++k) d[k] = (c->cx[k] * a + c->cy[k] * b) + c->g[k]; /* :41-43 */
    double n = norm3(d);
    for (int k = 0; k < 3; ++k) {
        ray[k] = (float)(c->pos[k] + d[k] * 140);                               /* :45 */
        ray[3 + k] = (float)(d[k] / n);                                         /* :46 */
    }
}

/* gen_rays(w,h,s) after np.random.seed(seed): rays = [6][N] float32 */
int oracle_gen_rays(uint32_t w, uint32_t h, uint32_t s, uint32_t seed, float *rays) {
    mt19937 m; mt_seed(&m, seed);
    camera c; camera_init(&c, w, h);
    const uint64_t N = (uint64_t)w * h * 4u * s;
    uint64_t p = 0;
    for (uint32_t i = 0; i < w; ++i) for (uint32_t j = 0; j < h; ++j)
        for (uint32_t sy = 0; sy < 2; ++sy) for (uint32_t sx = 0; sx < 2; ++sx)
            for (uint32_t k = 0; k < s; ++k, ++p) {
                double u1 = mt_double(&m), u2 = mt_double(&m);   /* r1 drawn before r2 :37,39 */
                float r[6];
                camera_ray(&c, w, h, i, j, sy, sx, u1, u2, r);
                for (int q = 0; q < 6; ++q) rays[(uint64_t)q * N + p] = r[q];
            }
    return 0;
}

/* gen_rays for a WINDOW of the np.random stream (frames whose whole stream is too long to walk: C3 has 1.7e10 paths):
 * state_in = the raw 624-word generator state whose tempering is output block first_block (624 words = the 4 words
 * of paths [156*first_block, 156*first_block+156)), or NULL to walk there from the seed.  Writes band-relative planes
 * rays[k*count + (p - first_path)] for paths [first_path, first_path+count), first_path >= 156*first_block.
 * state_out (or NULL): the raw state of block first_block (what a later call can pass as state_in).
 * state_end (or NULL): the raw state of block (first_path+count)/156, i.e. what the window that continues at path
 * first_path+count passes as state_in with that block as its first_block (windows chain without walking from the seed). */
int oracle_gen_rays_window(uint32_t w, uint32_t h, uint32_t s, uint32_t seed, const uint32_t *state_in, uint64_t first_block,
                           uint64_t first_path, uint64_t count, float *rays, uint32_t *state_out, uint32_t *state_end) {
    mt19937 m;
    if (state_in) { memcpy(m.mt, state_in, sizeof m.mt); m.idx = 0; }
    else {
        mt_seed(&m, seed);
        for (uint64_t b = 0; b <= first_block; ++b) { m.idx = 624; (void)mt_next(&m); } /* one twist per block */
        m.idx = 0;
    }
    if (state_out) memcpy(state_out, m.mt, sizeof m.mt);
    if (first_path < first_block * 156u) return 1;
    camera c; camera_init(&c, w, h);
    for (uint64_t skip = first_block * 156u; skip < first_path; ++skip) { (void)mt_double(&m); (void)mt_double(&m); }
    for (uint64_t q = 0; q < count; ++q) {
        const uint64_t p = first_path + q;
        uint64_t r = p / s;
        uint32_t sx = r & 1, sy = (r >> 1) & 1; r >>= 2;
        uint32_t j = (uint32_t)(r % h), i = (uint32_t)(r / h);
        double u1 = mt_double(&m), u2 = mt_double(&m);
        float ray[6];
        camera_ray(&c, w, h, i, j, sy, sx, u1, u2, ray);
        for (int k = 0; k < 6; ++k) rays[(uint64_t)k * count + q] = ray[k];
    }
    if (state_end) {
        if (m.idx >= 624) { m.idx = 624; (void)mt_next(&m); } /* exactly at a block boundary: the next block's state is one twist away */
        memcpy(state_end, m.mt, sizeof m.mt);
    }
    return 0;
}

/* the device-mode ray generator restated on the CPU: same camera maths, counter RNG */
int oracle_gen_rays_counter(const oracle_params *P, float *rays) {
    camera c; camera_init(&c, P->width, P->height);
    const uint64_t N = (uint64_t)P->width * P->height * 4u * P->samples;
    const uint64_t b = P->path_begin, n = P->path_count ? P->path_count : N - b;
    for (uint64_t p = b; p < b + n; ++p) {
        uint64_t k = p % P->samples, r = p / P->samples;
        uint32_t sx = r & 1, sy = (r >> 1) & 1; r >>= 2;
        uint32_t j = (uint32_t)(r % P->height), i = (uint32_t)(r / P->height);
        (void)k;
        double u1, u2; path_uniforms(P->seed, p, &u1, &u2);
        float ray[6];
        camera_ray(&c, P->width, P->height, i, j, sy, sx, u1, u2, ray);
        for (int q = 0; q < 6; ++q) rays[(uint64_t)q * N + p] = ray[q];
    }
    return 0;
}

/* gen_spheres(): gen_data.py:92-132.  out = 128 floats. */
int oracle_gen_spheres(float *out) {
    static const double tab[8][10] = {
        {1e5, 1e5 + 1, 40.8, 81.6, 0, 0, 0, 0.435, 0.376, 0.667},
        {1e5, -1e5 + 99, 40.8, 81.6, 0, 0, 0, 0.667, 0.129, 0.086},
        {1e5, 50, 40.8, 1e5, 0, 0, 0, 0.270, 0.725, 0.486},
        {1e5, 50, 40.8, -1e5 + 170, 0, 0, 0, 0, 0, 0},
        {1e5, 50, 1e5, 81.6, 0, 0, 0, 0.5, 0.5, 0.5},
        {1e5, 50, -1e5 + 81.6, 81.6, 0, 0, 0, 0.141, 0.408, 0.635},
        {16.5, 27, 16.5, 47, 0, 0, 0, 0.999, 0.999, 0.999},
        {600, 50, 681.6 - 0.27, 81.6, 12, 12, 12, 0, 0, 0}};
    memset(out, 0, 128 * sizeof(float));
    for (int k = 0; k < 8; ++k)
        for (int m = 0; m < 10; ++m) {
            double v = tab[k][m];
            if (m == 0) v = v * v; /* :109, squared in float64 before the float32 cast */
            out[m * 8 + k] = (float)v;
        }
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* numpy's pairwise float32 sum (np.mean over a strided run of n floats)                  */
static float pairwise_sum(const float *a, uint64_t n, uint64_t stride) {
    if (n < 8) {
        float r = 0.0f;
        for (uint64_t i = 0; i < n; ++i) r = r + a[i * stride];
        return r;
    } else if (n <= 128) {
        float r[8];
        uint64_t i;
        for (i = 0; i < 8; ++i) r[i] = a[i * stride];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] = r[j] + a[(i + j) * stride];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res = res + a[i * stride];
        return res;
    } else {
        uint64_t n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_sum(a, n2, stride) + pairwise_sum(a + n2 * stride, n - n2, stride);
    }
}

/* one pixel of decode_color (data_visualization.py:36-54) from the four sub-pixel means */
static double pixel_from_means(const float m[4]) {
    double acc = 0.0;
    for (int q = 0; q < 4; ++q) acc += (double)m[q];  /* :41-45 sum_color += np.mean(...) */
    return acc / 4;                                    /* :46 */
}

/* colors [3][N] -> f64 [W*H][3] (pre-clip, x-major q=i*H+j, NOT flipped), fb float32
 * [3][W*H] clipped, u8 [W*H][3].  Any output may be NULL. */
int oracle_decode_color(const float *colors, uint32_t w, uint32_t h, uint32_t s, double *pre, float *fb, uint8_t *u8) {
    const uint64_t npix = (uint64_t)w * h, N = npix * 4u * s;
    for (uint64_t q = 0; q < npix; ++q)
        for (int c = 0; c < 3; ++c) {
            float m[4];
            for (int sub = 0; sub < 4; ++sub)
                m[sub] = pairwise_sum(colors + (uint64_t)c * N + (q * 4 + sub) * s, s, 1) / (float)s; /* np.mean */
            double v = pixel_from_means(m);
            if (pre) pre[q * 3 + c] = v;
            double cl = v < 0 ? 0 : (v > 1 ? 1 : v);     /* :54 */
            if (fb) fb[(uint64_t)c * npix + q] = (float)cl;
            if (u8) u8[q * 3 + c] = (uint8_t)(cl * 255); /* :55-57 truncation */
        }
    return 0;
}

/* write_ppm (data_visualization.py:11-17).  The reference loops `for i in range(w): for j
 * in range(h): data[j, i]` on the (w,h,3) array returned by decode_color, where the second
 * index is already the flipped row: file row i <-> y = h-1-i, file column j <-> x = j.
 * That is only in range for w == h; the non-square definition used here is the evident
 * one: h rows of w pixels.  u8 = [W*H][3] x-major, not flipped. */
int oracle_write_ppm(const char *path, uint32_t w, uint32_t h, const uint8_t *u8) {
    FILE *f = fopen(path, "w");
    if (!f) return 1;
    fprintf(f, "P3\n%u %u\n255\n", w, h);
    for (uint32_t row = 0; row < h; ++row) {
        uint32_t y = h - 1 - row;
        for (uint32_t x = 0; x < w; ++x) {
            const uint8_t *p = u8 + ((uint64_t)x * h + y) * 3;
            fprintf(f, "%u %u %u ", p[0], p[1], p[2]);
        }
        fprintf(f, "\n");
    }
    fclose(f);
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* fused frame: pixels [pixel_begin, pixel_begin+pixel_count), outputs indexed from 0.   */
int oracle_render_frame(const oracle_params *P, const float *sph, uint64_t pixel_begin, uint64_t pixel_count,
                        float *fb, uint8_t *u8, double *pre, int threads, uint64_t *traced_out) {
    camera cam; camera_init(&cam, P->width, P->height);
    const uint32_t S = P->samples;
    uint64_t traced = 0;
    int bad = 0;
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads > 0 ? threads : 1) reduction(+ : traced)
#endif
    for (int64_t qq = 0; qq < (int64_t)pixel_count; ++qq) {
        const uint64_t q = pixel_begin + (uint64_t)qq;
        const uint32_t i = (uint32_t)(q / P->height), j = (uint32_t)(q % P->height);
        float *buf = (float *)malloc((size_t)S * 3 * sizeof(float));
        if (!buf) { bad = 1; continue; }
        float m[3][4];
        for (uint32_t sub = 0; sub < 4; ++sub) {
            const uint32_t sy = sub >> 1, sx = sub & 1;
            for (uint32_t k = 0; k < S; ++k) {
                const uint64_t p = (q * 4 + sub) * S + k;
                double u1, u2; path_uniforms(P->seed, p, &u1, &u2);
                float ray[6], out[3];
                camera_ray(&cam, P->width, P->height, i, j, sy, sx, u1, u2, ray);
                traced += trace_path(P, sph, p, ray[0], ray[1], ray[2], ray[3], ray[4], ray[5], out);
                buf[k] = out[0]; buf[S + k] = out[1]; buf[2 * S + k] = out[2];
            }
            for (int c = 0; c < 3; ++c) m[c][sub] = pairwise_sum(buf + (size_t)c * S, S, 1) / (float)S;
        }
        free(buf);
        for (int c = 0; c < 3; ++c) {
            double v = pixel_from_means(m[c]);
            if (pre) pre[(uint64_t)qq * 3 + c] = v;
            double cl = v < 0 ? 0 : (v > 1 ? 1 : v);
            if (fb) fb[(uint64_t)c * pixel_count + (uint64_t)qq] = (float)cl;
            if (u8) u8[(uint64_t)qq * 3 + c] = (uint8_t)(cl * 255);
        }
    }
    if (traced_out) *traced_out = traced;
    return bad;
}

/* build-defined large scene (BASELINE config 4): restated from the product's generator   */
/* so that tests can check the two agree.  [10][Ns] planes, zero padded to 128 floats.    */
size_t oracle_scene_floats(uint32_t ns) {
    size_t n = (size_t)ns * 10;
    return (n + 127) / 128 * 128;
}
int oracle_gen_scene(uint32_t ns, uint64_t seed, float *out) {
    if (ns < 8) return 1;
    float base[128];
    oracle_gen_spheres(base);
    memset(out, 0, oracle_scene_floats(ns) * sizeof(float));
    for (uint32_t k = 0; k < ns; ++k) {
        float rec[10];
        if (k < 6) for (int m = 0; m < 10; ++m) rec[m] = base[m * 8 + k];           /* walls */
        else if (k == ns - 1) for (int m = 0; m < 10; ++m) rec[m] = base[m * 8 + 7]; /* light */
        else {
            uint64_t s = splitmix64(seed ^ splitmix64(0x5CE7E000ull + k));
            if (s == 0) s = 0x9E3779B97F4A7C15ull;
            double u[7];
            for (int m = 0; m < 7; ++m) u[m] = (double)(xorshift64s(&s) >> 11) * (1.0 / 9007199254740992.0);
            double r = 0.5 + 1.5 * u[0];
            rec[0] = (float)(r * r);
            rec[1] = (float)(1.0 + 98.0 * u[1]);
            rec[2] = (float)(81.6 * u[2]);
            rec[3] = (float)(170.0 * u[3]);
            rec[4] = rec[5] = rec[6] = 0.0f;
            rec[7] = (float)(0.1 + 0.899 * u[4]);
            rec[8] = (float)(0.1 + 0.899 * u[5]);
            rec[9] = (float)(0.1 + 0.899 * u[6]);
        }
        for (int m = 0; m < 10; ++m) out[(size_t)m * ns + k] = rec[m];
    }
    return 0;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
