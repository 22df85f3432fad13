/*
 * render_mi355x.h -- C-ABI of librender_mi355x.so, the MI355X (gfx950) drop-in for the
 * hot path of KVM-Explorer/AscendPathTracing (reference paths are relative to the
 * reference repository root):
 *
 *     ray-generate -> ray/sphere intersect -> mirror-reflect / throughput loop -> colour
 *
 * Every entry point below names the reference interface it replaces.  Plain pointers and
 * sizes only; no C++ or torch types.  All device entry points ENQUEUE work on the HIP
 * stream they are given and return without synchronising (the caller synchronises, as
 * src/main.cpp:75 does with aclrtSynchronizeStream); they never allocate, free or retain
 * memory, so they are legal inside a hipGraph capture.  (The exceptions say so: apt_render_host and
 * apt_multi_* are synchronous conveniences that own device buffers.)
 *
 * Thread safety.  apt_last_error() / apt_last_status() are PER THREAD and describe the last call made on
 * that thread: every entry point clears the record on entry and sets it on failure.  The settings the
 * reference keeps as compile-time constants live in an apt_context; its setters and the snapshot a render
 * call takes are serialised inside the context, so concurrent calls on one context are safe and each call
 * sees one consistent set of values; different contexts share nothing.  The context-free forms
 * (render_do, apt_set_default_params, apt_set_trace_counter, apt_set_refill_lanes) act on one process-wide
 * default context with the same guarantees -- two threads that want DIFFERENT render_do parameters use two
 * contexts.
 *
 * Buffers (little-endian IEEE float32):
 *   rays     [6][N]   planes ox,oy,oz,dx,dy,dz              scripts/gen_data.py:65-71
 *   spheres  [10][Ns] planes r^2,x,y,z,emX,emY,emZ,colX,colY,colZ, zero padded to a
 *                     multiple of 512 bytes                  scripts/gen_data.py:106-127
 *   colors   [3][N]   planes r,g,b                           src/render.cpp:218-220
 *   path index p = (((i*H + j)*2 + sy)*2 + sx)*S + k         scripts/gen_data.py:32-36
 *   N = W*H*4*S.
 */
#ifndef RENDER_MI355X_H
#define RENDER_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* librender_mi355x.so is built with -fvisibility=hidden and an export list (csrc/apt_exports.map): what this header declares --
 * and the C++-mangled render_do of src/main.cpp:9-10 -- is all it exports. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define APT_ABI_VERSION 3   /* additive since 3 (round 5): apt_grid_flags, apt_context_get_debug / apt_get_debug, APT_FLAG_GRID_SLOTS, APT_DEV_GRID_MISMATCH */

/* status codes returned by the *_ex / frame entry points (render_do itself is void,
 * like the reference, and reports through apt_last_status() / apt_last_error()). */
enum {
    APT_OK = 0,
    APT_ERR_ARG = 1,       /* null pointer, zero size, size not representable            */
    APT_ERR_STRUCT = 2,    /* struct_size does not match this library                    */
    APT_ERR_SCENE = 3,     /* num_spheres == 0 or light_index out of range               */
    APT_ERR_DEVICE = 4,    /* HIP runtime error (no device, launch failure, ...), or a kernel  */
                           /* reported a failure through the device status word (below)   */
    APT_ERR_IO = 5         /* file contract helpers                                      */
};

/* arithmetic modes */
enum {
    APT_MODE_KERNEL = 0,   /* K-mode: operation order of src/rt_helper.h, all fp32       */
    APT_MODE_ORACLE = 1    /* O-mode: scripts/gen_data.py:246-429 test_soa; the two dot  */
                           /* products of the shading step accumulate in float64 and an  */
                           /* all-miss selects sphere index -1 (Python wrap-around)      */
};

/* flags */
enum {
    APT_FLAG_RETIRE = 1u,  /* result-preserving retirement of finished paths: a path     */
                           /* whose alive bit is cleared or whose throughput is (0,0,0)   */
                           /* stops bouncing; colours are bit-identical either way.       */
    APT_FLAG_BAND_BUFFERS = 8u, /* render_do_ex / apt_gen_rays_device / apt_gen_rays_mt_device_ex: `rays` and  */
                           /* `colors` hold ONLY paths [path_begin, path_begin+path_count): planes of      */
                           /* path_count floats, element (plane k, path p) at [k*path_count + p-path_begin] */
                           /* (default: full [6][N] / [3][N] buffers indexed by the absolute path).  With    */
                           /* it the reference's exact pipeline runs band by band in a bounded buffer.       */
    APT_FLAG_EMISSION = 4u,/* colour = throughput * emission(light sphere) per channel       */
                           /* (spheres.bin planes 4..6) instead of the literal gain 12 of      */
                           /* render.cpp:194-196; identical on the reference scene (em = 12).  */
    APT_FLAG_GRID_SLOTS = 16u, /* render_frame with `accel`: the CALLER states that the grid at `accel` was built for this scene and    */
                           /* carries the pair-slot tables (apt_grid_flags() says so for a built grid).  The frame then takes ONE     */
                           /* launch -- the sample-queue kernel's grid form -- instead of two (without the flag both frame kernels are  */
                           /* launched and the grid's own header decides ON THE DEVICE which one renders; the other returns at once).   */
                           /* A grid that does not keep the promise is not walked: the kernel renders nothing and reports              */
                           /* APT_DEV_GRID_MISMATCH through the device status word.  Same image either way.                            */
    APT_FLAG_RR = 2u       /* EXTENSION (not in the reference; BASELINE config 5): Russian */
                           /* roulette.  After shading bounce d (0-based) with d+1 >=       */
                           /* rr_start, a path that is alive with q = max(r,g,b) > 0        */
                           /* survives with probability p = clamp(q, 0.05, 0.95): u =       */
                           /* 24-bit uniform from splitmix64(key + (d+1)*0x9E3779B97F4A7C15) */
                           /* >> 40, key = splitmix64(seed ^ splitmix64(path index)); u >= p */
                           /* zeroes the throughput, otherwise it is multiplied by 1/p.      */
                           /* Unbiased; changes the image by noise only.  Same fp32 ops on   */
                           /* GPU and in the CPU restatement (bitwise equal).                */
};

/* Run-time form of the reference's compile-time constants
 * (src/common.h:4-14, src/render.cpp:141,194-196, scripts/gen_data.py:6-10). */
typedef struct apt_render_params {
    uint32_t struct_size;   /* = sizeof(apt_render_params)                               */
    uint32_t width;         /* WIDTH   (common.h:4)                                      */
    uint32_t height;        /* HEIGHT  (common.h:5)                                      */
    uint32_t samples;       /* SAMPLES (common.h:6); samples per pixel = 4*samples       */
    uint32_t depth;         /* bounce count, literal 5 at render.cpp:141                 */
    uint32_t num_spheres;   /* SPHERE_NUM (common.h:10)                                  */
    int32_t  light_index;   /* literal 7 at rt_helper.h:776 / gen_data.py:381            */
    float    eps;           /* EPSILON (common.h:9)                                      */
    float    gain;          /* literal 12 at render.cpp:194-196                          */
    uint32_t mode;          /* APT_MODE_*                                                */
    uint32_t flags;         /* APT_FLAG_*                                                */
    uint32_t rr_start;      /* APT_FLAG_RR: first bounce count at which roulette applies; 0 = 3 */
    uint64_t path_begin;    /* first path index this call renders (multi-GPU shard)      */
    uint64_t path_count;    /* number of paths this call renders; 0 = all N              */
    uint64_t seed;          /* device ray generation only                                */
    uint64_t accel;         /* DEVICE address of a grid built by apt_build_grid_host, or 0 */
} apt_render_params;

/* Fill *p with the reference defaults: 16x16, samples 1, depth 5, 8 spheres, light 7,
 * eps 1e-4, gain 12, K-mode, no flags, whole image. */
void apt_default_params(apt_render_params *p);

/* ---- the reference boundary ---------------------------------------------------------
 * Replaces  void render_do(uint32_t blockDim, void *l2ctrl, void *stream,
 *                          uint8_t *rays, uint8_t *spheres, uint8_t *colors)
 * declared at src/main.cpp:9-10, defined at src/render.cpp:262-266 (body:
 * render<<<blockDim, l2ctrl, stream>>>(rays, spheres, colors)).
 *   blockDim  number of contiguous partitions in the reference (8); the result does not
 *             depend on it here, it is accepted and ignored
 *   l2ctrl    ignored
 *   stream    hipStream_t (0 = the default stream)
 *   rays, spheres, colors   DEVICE pointers, sizes 24*N, 512 and 12*N bytes
 *                           (src/main.cpp:47-49) with the reference's compile-time
 *                           W=H=16, S=1, depth 5; override with apt_set_default_params.
 * Asynchronous; errors are recorded for apt_last_error(). */
void render_do(uint32_t blockDim, void *l2ctrl, void *stream,
               uint8_t *rays, uint8_t *spheres, uint8_t *colors);

/* The same entry under a second name.  The reference declares render_do with C++ linkage
 * (`extern void render_do(...)` without extern "C", src/main.cpp:9-10), so an UNMODIFIED main.o references
 * the mangled symbol _Z9render_dojPvS_PhS0_S0_: the library exports that symbol as well (render_do_cxx.cpp,
 * which forwards here), and a translation unit holding the reference's declaration verbatim links against
 * librender_mi355x.so as it is (tests/test_host_abi.py builds one). */
void apt_render_do(uint32_t blockDim, void *l2ctrl, void *stream,
                   uint8_t *rays, uint8_t *spheres, uint8_t *colors);

/* The parameters render_do() uses (the process-wide default context).  Returns APT_OK or APT_ERR_*. */
int apt_set_default_params(const apt_render_params *p);

/* The CPU-simulator shape of the boundary: src/main.cpp:21-44 calls ICPU_RUN_KF(render, blockDim, rays,
 * spheres, colors) on HOST buffers and the call is synchronous.  This entry takes the same HOST buffers
 * (24*N, 512 (padded table) and 12*N bytes for the default context's parameters), copies them to device 0's
 * current device, renders, copies the colours back and returns when they are there.  Allocates and frees its
 * device buffers; not capture-safe.  Whole frames only, like the reference's call: APT_ERR_ARG when the default
 * parameters carry a strict path sub-range (path_begin != 0, or a path_count that is neither 0 nor N).  Checks the device
 * status word before it returns.  (The arithmetic still runs on the GPU: there is no CPU path.) */
int apt_render_host(uint32_t blockDim, const uint8_t *rays, const uint8_t *spheres, uint8_t *colors);

/* ---- contexts: per-caller settings instead of process-wide ones ---------------------------------
 * A context holds what the reference fixes at compile time (the parameters its render_do renders with) and
 * the diagnostics knobs below.  See "Thread safety" at the top. */
typedef struct apt_context apt_context;
apt_context *apt_context_create(void);                 /* reference defaults; NULL when out of memory */
void apt_context_destroy(apt_context *ctx);
int  apt_context_set_params(apt_context *ctx, const apt_render_params *p);
int  apt_context_set_trace_counter(apt_context *ctx, uint64_t *device_counter);
int  apt_context_set_refill_lanes(apt_context *ctx, uint32_t lanes);

/* Device-side failure channel (the reference asserts INSIDE its kernel: src/render.cpp:68-73 DataFormatCheck / ASSERT).  Every context
 * owns one uint32 status word per device it has launched on; a kernel that has to give up ORs a bit into it (APT_DEV_*), and the frame
 * it wrote is then incomplete.  The word is STICKY: it keeps every failure since the last check.
 *   apt_context_check(ctx, stream)  waits for `stream` (synchronising; not capture-safe), reads the word of the current device and
 *       clears it: APT_OK, or APT_ERR_DEVICE with the bits named in apt_last_error().  apt_check(stream) = the default context.
 *   apt_render_host and apt_multi_render check it themselves before they return.
 * The word is allocated on a context's first launch on a device -- through ANY launching entry point, render_do / render_do_ex /
 * render_frame / apt_render_frame_mt included: the one allocation those otherwise allocation-free calls ever make -- (one hipMalloc of
 * 4 bytes, once per context and device, device indices 0..15; a device beyond that runs without a word; skipped while `stream` is being
 * captured -- a context whose FIRST launch on a device happens inside a graph capture runs without the word until a launch or an
 * apt_context_check outside a capture has made it). */
enum {
    APT_DEV_QUEUE_GUARD = 1u,  /* sample-queue kernel (APT_FLAG_RETIRE, 8 spheres): the bound on a wave's loop turns ran out      */
    APT_DEV_GRID_TURNS = 2u,   /* sample-queue kernel, grid form: the bound on a wave's walk turns ran out                         */
    APT_DEV_LDS_BASE = 4u,     /* sample-queue kernels: the dynamic LDS region does not start at LDS address 0 (a build problem)    */
    APT_DEV_GRID_MISMATCH = 8u /* APT_FLAG_GRID_SLOTS: the grid at `accel` is not this scene's or has no pair-slot tables           */
};
int  apt_context_check(apt_context *ctx, void *stream);
int  apt_check(void *stream);

/* Measurement knobs of a context (0 = the library's own choice); tests and profiling scripts select kernels through these, never
 * through the process environment (the APT_* variables of earlier rounds are read once, when a context is created, as initial values):
 *   "queue_ppw" 1..4096 pixels per wave of the sample-queue kernels    "queue_nbuf" 2..16 colour buffers    "queue_lds_pad" bytes
 *   "grid_walk" 1 = frames of a scene behind a grid use render_frame_kernel's nested item walk only (bit-identical, slower)
 *   "grid_spheres_per_cell" cell size of apt_build_grid_host / apt_build_grid_device (default context's value)
 * APT_ERR_ARG for an unknown key or a value out of range.  apt_set_debug = the default context.
 * apt_context_get_debug / apt_get_debug read a knob's current value (so that a caller can restore what it found). */
int  apt_context_set_debug(apt_context *ctx, const char *key, double value);
int  apt_set_debug(const char *key, double value);
int  apt_context_get_debug(apt_context *ctx, const char *key, double *value);
int  apt_get_debug(const char *key, double *value);
void apt_context_render_do(apt_context *ctx, uint32_t blockDim, void *l2ctrl, void *stream,
                           uint8_t *rays, uint8_t *spheres, uint8_t *colors);
int  apt_context_render_do_ex(apt_context *ctx, const apt_render_params *p, void *stream,
                              const float *rays, const float *spheres, float *colors);
int  apt_context_render_frame(apt_context *ctx, const apt_render_params *p, void *stream, const float *spheres,
                              uint64_t pixel_begin, uint64_t pixel_count, float *fb, uint8_t *fb_u8);

/* Extended form of the same boundary with run-time parameters.
 * rays/colors are the FULL [6][N] / [3][N] device buffers; the call reads and writes only
 * paths [path_begin, path_begin+path_count). */
int render_do_ex(const apt_render_params *p, void *stream,
                 const float *rays, const float *spheres, float *colors);

/* ---- rays generated on the device, samples accumulated on the device ----------------
 * Fuses gen_rays (scripts/gen_data.py:21-75, camera maths in float64; the two uniforms of path p
 * are outputs 2p+1 and 2p+2 of the SplitMix64 stream seeded with splitmix64(p->seed) -- 52 high
 * bits each -- instead of the host MT19937 stream), the render loop
 * (src/render.cpp:104-207) and decode_color (scripts/data_visualization.py:20-59: mean
 * over S, sum over the 2x2 sub-pixels in float64, /4, clip, *255 truncation).
 *   pixel_begin, pixel_count   range of x-major pixel indices q = i*H + j rendered
 *   fb      device, float32 [3][pixel_count]: clipped pixel value in [0,1], planes r,g,b,
 *           indexed by q - pixel_begin (y NOT flipped; apt_write_ppm flips)
 *   fb_u8   device, uint8 [pixel_count][3] = trunc(clip * 255) from the float64 value,
 *           or NULL
 * p->path_begin/path_count are ignored here. */
int render_frame(const apt_render_params *p, void *stream, const float *spheres,
                 uint64_t pixel_begin, uint64_t pixel_count, float *fb, uint8_t *fb_u8);

/* ---- one process, several GPUs (the reference's 8-block split, src/render.cpp:9-10,24-27, across devices) ----
 * The frame's x-major pixel range is cut into num_bands*stripes contiguous stripes; band b renders stripes
 * b, b+num_bands, ... (stripes == 1: one contiguous band per device, the reference's split; stripes > 1
 * interleaves them, which balances APT_FLAG_RETIRE's uneven columns) on device_ids[b] with render_frame on a
 * stream of its own, and every stripe is copied straight into the full frame on the root device
 * (device_ids[0]) with hipMemcpyPeerAsync -- over xGMI each peer has its own link to the root, so the copies
 * of different bands run in parallel; a gather to one root needs no collective.  A device may appear more
 * than once in device_ids (several bands on one GPU).
 *   apt_multi_create   allocates per-band streams, the scene copy and stripe buffers (spheres_host: HOST table)
 *   apt_multi_render   renders one frame into fb_root [3][W*H] (+ u8_root [W*H][3] or NULL), DEVICE pointers on
 *                      the root device; synchronous (returns when the frame is complete);
 *                      band_kernel_ms: optional HOST float[num_bands], each band's kernel time (HIP events)
 * apt_render_params.accel is refused (a grid address belongs to one device).  One process per GPU with
 * torch.distributed (ascendpathtracing_amd/dist.py) is the other, equivalent way to shard. */
typedef struct apt_multi apt_multi;
int  apt_multi_create(const int *device_ids, uint32_t num_bands, uint32_t stripes, const apt_render_params *p,
                      const float *spheres_host, apt_multi **out);
int  apt_multi_render(apt_multi *m, float *fb_root, uint8_t *u8_root, float *band_kernel_ms);
void apt_multi_destroy(apt_multi *m);

/* First-hit debug mode, scripts/gen_data.py:134-188 test_scene: out [3][N] = emission of the
 * light sphere when it is the nearest hit, the hit sphere's colour otherwise, 0 on a miss
 * (test_scene's own arithmetic: float64-accumulated dot products). */
int apt_test_scene(const apt_render_params *p, void *stream, const float *rays,
                   const float *spheres, float *out);

/* Device gen_rays with the counter-based generator: writes rays [6][N] planes for paths
 * [path_begin, path_begin+path_count) of the full buffer (same rays render_frame traces). */
int apt_gen_rays_device(const apt_render_params *p, void *stream, float *rays);

/* Device gen_rays that is BIT-EXACT with scripts/gen_data.py:21-75 under np.random.seed(seed):
 * the MT19937 stream is regenerated on the device from host-made checkpoints (the raw 624-word
 * state every `stride` output blocks; one block = 624 words = 156 paths).
 *   apt_mt19937_checkpoints_host(seed, num_blocks, stride, states): states = HOST uint32
 *       [ceil(num_blocks/stride)][624]; num_blocks = ceil(N/156).  Sequential, done once per seed;
 *       a table made for a longer stream serves every shorter one.
 *   apt_gen_rays_mt_device(p, stream, checkpoints_dev, stride, num_checkpoints, rays): rays [6][N],
 *       paths [path_begin, path_begin+path_count).  p->seed is not used (the seed is in the table). */
int apt_mt19937_checkpoints_host(uint32_t seed, uint64_t num_blocks, uint32_t stride, uint32_t *states);
int apt_gen_rays_mt_device(const apt_render_params *p, void *stream, const uint32_t *checkpoints,
                           uint32_t stride, uint64_t num_checkpoints, float *rays);
/* Windowed forms, for frames whose whole stream is too long to tabulate at once (C3: 1.7e10 paths):
 *   apt_mt19937_checkpoints_window(state_in, seed, first_block, num_blocks, stride, states, state_out):
 *       checkpoints of output blocks first_block + k*stride, k < ceil(num_blocks/stride).  state_in = the raw
 *       624-word state of block first_block (from an earlier call's state_out, or a stored one), or NULL to twist
 *       there from the seed (first_block+1 sequential twists).  state_out (or NULL) receives the raw state of
 *       block first_block + num_blocks, so that windows chain.
 *   apt_gen_rays_mt_device_ex: as apt_gen_rays_mt_device with a table whose entry 0 is block first_block;
 *       the call's path range must lie inside the window; honours APT_FLAG_BAND_BUFFERS. */
int apt_mt19937_checkpoints_window(const uint32_t *state_in, uint32_t seed, uint64_t first_block, uint64_t num_blocks,
                                   uint32_t stride, uint32_t *states, uint32_t *state_out);
int apt_gen_rays_mt_device_ex(const apt_render_params *p, void *stream, const uint32_t *checkpoints, uint32_t stride,
                              uint64_t num_checkpoints, uint64_t first_block, float *rays);

/* The reference's EXACT pipeline in one launch (round 3; SURVEY 8(f)-1 "removes the rays.bin file/HBM boundary"):
 * MT19937 gen_rays (scripts/gen_data.py:21-75 under np.random.seed, :438) -> the render loop (src/render.cpp:104-207; in
 * APT_MODE_ORACLE the arithmetic of gen_data.py:246-429 test_soa) -> decode_color (scripts/data_visualization.py:36-57),
 * without the [6][N] ray buffer and the [3][N] colour buffer: bit-identical to apt_gen_rays_mt_device -> render_do_ex ->
 * apt_decode_color_device.  One workgroup per GROUP of 78 pixels (78 * 4 * samples paths = 2 * samples generator blocks of
 * 156 paths, for EVERY sample count): checkpoints[k] = the raw 624-word state of block (first_group + k) * 2 * samples,
 * i.e. apt_mt19937_checkpoints_window(state_in, seed, first_group * 2 * samples, groups * 2 * samples, 2 * samples, ...).  Renders
 * pixels [pixel_begin, pixel_begin + pixel_count) into fb [3][pixel_count] (+ fb_u8 [pixel_count][3] or NULL); the table must
 * cover groups pixel_begin / 78 .. (pixel_begin + pixel_count - 1) / 78.  The 8-sphere scene, no APT_FLAG_RR; any sample count that has a
 * pairwise-sum plan (every count <= 7688; since round 4 -- before: 8, 16, ..., 256 only -- incl. the reference's own default, 16 x 16 with
 * samples = 1, src/common.h:4-6).  Enqueues on `stream`, allocates nothing. */
int apt_render_frame_mt(const apt_render_params *p, void *stream, const uint32_t *checkpoints, uint64_t num_checkpoints,
                        uint64_t first_group, const float *spheres, uint64_t pixel_begin, uint64_t pixel_count,
                        float *fb, uint8_t *fb_u8);

/* Device decode_color: colors [3][N] -> fb float32 [3][W*H] (+ fb_u8 [W*H][3] or NULL),
 * same arithmetic as scripts/data_visualization.py:20-59. */
int apt_decode_color_device(const apt_render_params *p, void *stream, const float *colors,
                            float *fb, uint8_t *fb_u8);

/* The same for a band: colors_band = float32 [3][pixel_count*4*S] holding the paths of pixel_count consecutive
 * pixels (APT_FLAG_BAND_BUFFERS layout) -> fb [3][pixel_count] (+ fb_u8 [pixel_count][3] or NULL).  decode_color
 * does not depend on where in the image the pixels lie. */
int apt_decode_color_band(const apt_render_params *p, void *stream, const float *colors_band, uint64_t pixel_count,
                          float *fb, uint8_t *fb_u8);

/* ---- host-side helpers (no GPU) -------------------------------------------------------
 * Host gen_rays, bit-exact with scripts/gen_data.py:21-75 under np.random.seed(seed)
 * (MT19937 legacy stream).  rays = HOST float32 [6][N]. */
int apt_gen_rays_host(uint32_t width, uint32_t height, uint32_t samples, uint32_t seed,
                      float *rays);

/* Host gen_spheres (scripts/gen_data.py:92-132): writes the 128-float / 512-byte table. */
int apt_gen_spheres_host(float *spheres128);

/* Build-defined large scene (BASELINE config 4; the reference has no generator):
 * spheres 0..5 the six walls, 6..Ns-2 random small spheres, Ns-1 the light.
 * Writes [10][Ns] planes zero padded to a multiple of 128 floats; *out_floats receives the
 * padded length.  Pass spheres == NULL to query the length. */
int apt_gen_scene_host(uint32_t num_spheres, uint64_t seed, float *spheres,
                       size_t *out_floats);

/* Acceleration structure for scenes far larger than LDS (num_spheres != 8): a uniform grid over the
 * small spheres plus an always-tested list of the large ones (walls, light).  Built on the HOST from
 * the same [10][Ns] table the kernels read; the caller copies the `*out_bytes` bytes to the device and
 * passes that address in apt_render_params.accel.  Pass grid == NULL to query the size.  Rendering
 * with it is bit-identical to rendering without: every candidate goes through the reference's exact
 * intersection arithmetic, the traversal only skips spheres that provably cannot be hit, and ties keep
 * the lowest sphere index.  The buffer holds two forms of the lists: item ranges (the nested walk of buffer
 * mode and of samples < 8) and, since round 3, pair-slot tables (two candidates per 32-byte slot, one word per
 * cell) for the frame kernel's per-lane walk; apt_render_frame picks the kernel from what the buffer carries,
 * on the device. */
int apt_build_grid_host(const float *spheres_host, uint32_t num_spheres, void *grid, size_t *out_bytes);

/* The flags a built grid earns for a scene of `num_spheres` spheres, from the first 128 bytes of the buffer (HOST memory: the
 * buffer apt_build_grid_host filled, or a copy of the head of a device-built one): APT_FLAG_GRID_SLOTS when it is a grid for that
 * sphere count with pair-slot tables, else 0.  OR the result into apt_render_params.flags next to `accel`. */
uint32_t apt_grid_flags(const void *grid_head_host, uint32_t num_spheres);

/* The same grid built ON THE DEVICE from the [10][Ns] table in device memory: byte-identical to what
 * apt_build_grid_host writes for that scene (radix-select median, scans and atomics in kernels; the scalar header
 * arithmetic is the host's own).  grid_dev = DEVICE buffer of `capacity` bytes, or NULL to query the size
 * (*out_bytes).  Synchronous on `stream` (two small read-backs size the buffer) and allocates its workspace: a
 * build step, not capture-safe.  Up to 512 cells per axis (round 1: 128). */
int apt_build_grid_device(const float *spheres_dev, uint32_t num_spheres, void *stream, void *grid_dev,
                          size_t capacity, size_t *out_bytes);

/* P3 writer of scripts/data_visualization.py:11-17 from a [pixel][3] uint8 image in
 * x-major pixel order (q = i*H + j, y not flipped). */
int apt_write_ppm(const char *path, uint32_t width, uint32_t height, const uint8_t *fb_u8);

/* ---- diagnostics ------------------------------------------------------------------- */
/* Optional DEVICE uint64[4] statistics block.  [0] += ray segments actually traced by every
 * render launch (== paths*depth unless APT_FLAG_RETIRE); [1], [2] += lane-slots (64 per wave-level
 * execution) spent in bounce / ray-generate by the refill loop of render_frame; [3] += wave-level
 * exact re-runs of a bounce (full-trace Ns = 8 loop).  NULL disables it.
 * The caller zeroes it; process-wide. */
int apt_set_trace_counter(uint64_t *device_counter);

/* Self-test of the hot loop's fast correctly-rounded sqrt: compares it with sqrtf() for every
 * float whose bit pattern lies in [first_bits, first_bits+count) (count = 2^32 covers all).
 * variant 0 = neighbour check (shade), 1..3 = FMA-correction candidates; the intersect loop uses 2.
 * device_result2[0] += number of mismatches, device_result2[1] = min(first mismatching bits);
 * the caller initialises them to 0 and ~0. */
int apt_selftest_sqrt(int variant, void *stream, uint64_t first_bits, uint64_t count,
                      uint64_t *device_result2);

/* Self-test of the shared-reciprocal divide of the shading step against the plain `/` on
 * `count` hashed operand sets starting at counter `first`.  device_result3[0] += mismatches,
 * [1] = min(first mismatching counter), [2] += sets inside the accepted operand range.
 * Initialise to 0, ~0, 0. */
int apt_selftest_div3(void *stream, uint64_t first, uint64_t count, uint64_t *device_result3);

/* Tuning knob of the APT_FLAG_RETIRE compaction in render_frame: a wave runs its (expensive,
 * float64) ray-generate when at least `lanes` of its 64 lanes have an empty ray slot (default
 * 32).  Speed only: results are bit-identical for every value.  Default context. */
int apt_set_refill_lanes(uint32_t lanes);

int         apt_abi_version(void);
const char *apt_last_error(void);       /* this thread's last call: "" when it succeeded */
int         apt_last_status(void);      /* this thread's last call: APT_OK or APT_ERR_* (how a caller of the void
                                           render_do learns the outcome) */
int         apt_device_count(void);     /* number of HIP devices, 0 when none */

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* RENDER_MI355X_H */
