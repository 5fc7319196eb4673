/*
 * rtiow_hip.h -- C ABI of librtiow_hip.so, the MI355X (gfx950) replacement for
 * the per-pixel Monte-Carlo loop of Druthyn/rtiow.
 *
 * What it replaces.  The reference has no FFI or plugin seam: the path is the
 * inlined iterator expression at src/main.rs:122-139 and everything it calls
 * (ray_color main.rs:38-57, HittableList::hit shapes/mod.rs:54-70, Sphere::hit
 * shapes/sphere.rs:15-41, {Lambertian,Metal,Dialectric}::scatter
 * materials.rs:21-31,48-62,76-105, Camera::get_ray camera.rs:47-54 and
 * Color::to_rgba vec3.rs:403-421).  Each entry point below names the reference
 * lines it stands in for; INTEGRATION.md shows the Rust `extern "C"` block and
 * the edit to main.rs that binds them.
 *
 * Rules of the boundary
 *   - plain C types, plain pointers and sizes; no C++/torch types;
 *   - every function returns 0 on success or a negative rt_status; it never
 *     aborts or throws across the ABI (the reference unwrap()s, main.rs:147,177);
 *     rt_last_error() gives the thread-local message of the last failure;
 *   - the caller owns every buffer it passes; the library owns the device
 *     memory tied to an rt_context;
 *   - a context is used from one host thread at a time; it may have TWO
 *     renders in flight (rt_render_device on two different streams: the work
 *     counter, statistics words and events exist twice and are used in turn --
 *     see rt_render_device); distinct contexts (one per GPU) may be used
 *     concurrently;
 *   - there is NO CPU fallback: without a usable gfx950 device rt_create fails.
 *
 * Arithmetic.  Results are those of the reference's own f64 arithmetic: every
 * value that decides or shades a path is computed in IEEE binary64 in the
 * reference's operation order (no fused multiply-add), so a render equals the
 * literal CPU restatement (oracle/oracle_f64.c, Oracle B) bit for bit.  f32
 * is used only as a CONSERVATIVE FILTER in the sphere scan (it may send a
 * sphere to the exact f64 test needlessly, never the reverse; DESIGN.md
 * section 5.2).  A pixel's sum over samples is exact: every sample's radiance
 * is truncated to a 2^-32 grid and summed in an unsigned 64-bit integer, so
 * the result does not depend on how samples are sharded over lanes, launches,
 * passes or GPUs.
 */
#ifndef RTIOW_HIP_H
#define RTIOW_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTIOW_HIP_ABI_VERSION 5

typedef struct rt_context rt_context;

typedef enum {
    RT_OK = 0,
    RT_ERR_INVALID_ARGUMENT = -1,
    RT_ERR_NO_DEVICE = -2,       /* no HIP device / not gfx950 */
    RT_ERR_HIP = -3,             /* a HIP runtime call failed  */
    RT_ERR_NO_SCENE = -4,        /* render before rt_upload_scene */
    RT_ERR_OUT_OF_MEMORY = -5
} rt_status;

/* Material kinds: the three `impl Scatter` of materials.rs. */
enum { RT_LAMBERTIAN = 0, RT_METAL = 1, RT_DIALECTRIC = 2 };

/* One sphere, flattened, in the reference's own f64.  The reference keeps
 * Sphere{center,radius,mat} (shapes/sphere.rs:9-13) and the material structs
 * (materials.rs:9-11,34-37,64-66) private behind Box<dyn Hit>/Arc<dyn Scatter>,
 * so the host flattens each object where it is pushed (main.rs:64,87,93-99).
 * LIST ORDER IS PART OF THE INPUT: on equal t the later sphere wins
 * (mod.rs:61-67, sphere.rs:29). */
typedef struct {
    double center[3];
    double radius;
    double albedo[3];    /* Lambertian, Metal                        */
    double param;        /* Metal: fuzz; Dialectric: ir              */
    int32_t kind;        /* RT_LAMBERTIAN / RT_METAL / RT_DIALECTRIC */
    int32_t reserved;    /* 0                                        */
} rt_sphere;             /* 72 bytes */

/* Camera (camera.rs:4-13) minus `w`, which get_ray never reads; f64 as in the
 * reference.  The host computes it with its Camera::new (camera.rs:17-45). */
typedef struct {
    double origin[3];
    double lower_left_corner[3];
    double horizontal[3];
    double vertical[3];
    double u[3];
    double v[3];
    double lens_radius;
} rt_camera;             /* 152 bytes */

/* What main.rs:24-28,44 fixes at compile time, plus sharding.
 *
 * Rows: j = 0 is the BOTTOM image row, as in main.rs:122-132.  The image is
 * cut into row tiles of `tile_rows` rows; a call renders the tiles t with
 * t % shard_count == shard_index, and its output holds those rows only,
 * ascending j ("compact rows"; rt_shard_rows() gives how many).
 * shard_count = 1 renders the whole image. */
typedef struct {
    int32_t width, height;       /* main.rs:25-26 (>= 2: u,v divide by W-1,H-1)   */
    int32_t spp;                 /* samples per pixel rendered by this call :27    */
    int32_t sample_begin;        /* index of the first sample (additive passes)    */
    int32_t max_depth;           /* :28, 50                                        */
    double  t_min;               /* :44, 1e-4; must be > 0                         */
    uint64_t seed;               /* Philox key                                     */
    int32_t tile_rows;           /* >= 1                                           */
    int32_t shard_index;         /* 0 <= shard_index < shard_count                 */
    int32_t shard_count;         /* >= 1                                           */
    uint32_t flags;              /* RT_FLAG_*                                      */
} rt_params;

#define RT_FLAG_ACCUMULATE 0x1u  /* rt_render_device: add to d_fix instead of overwriting it */
#define RT_FLAG_NO_FILTER  0x2u  /* validation: send EVERY sphere to the exact f64 test     */
#define RT_FLAG_DIAG_STATS 0x4u  /* also fill rt_stats.candidates / exact_roots / live_per_bounce (~15 % slower) */
#define RT_FLAG_UNIFORM53  0x8u  /* every uniform from TWO Philox words, u = ((w0 << 32 | w1) >> 11) * 2^-53: the 53 random bits of
                                    rand's gen::<f64>() (main.rs:131-132, vec3.rs:31-33,63, materials.rs:96) instead of one word's 32
                                    (the default: u = w * 2^-32, symmetric ranges (int32_t)w * 2^-31; ABI <= 4: 24 bits); same draw
                                    order; another (equally valid) random stream, so frames differ from the default's; scan mode 5 or
                                    RT_FLAG_NO_FILTER, not with RT_FLAG_DIAG_STATS */
#define RT_FLAG_OVERLAPPED 0x10u /* rt_render_device: this launch is one of a sequence of passes that OVERLAP on two streams (below): the next
                                    pass fills its end-of-launch tail, so it takes the work blocks of 1 024 pixel-samples whatever its size
                                    (alone, a launch of < 2 x 10^8 pixel-samples is up to 6 % slower on them: its last blocks are its tail).
                                    The same frame either way; 2 x 250 spp at 1200x675: 1.045 -> 1.012 x one 500-spp launch */
#define RT_FLAG_KNOWN      0x1fu /* every other bit of rt_params.flags is an error (RT_ERR_INVALID_ARGUMENT), not ignored */

typedef struct {
    uint64_t samples;            /* pixel-samples finished                          */
    uint64_t rays_traced;        /* HittableList::hit calls (mod.rs:56)             */
    uint64_t sphere_tests;       /* rays_traced * n_spheres (sphere.rs:16 calls)    */
    uint64_t candidates;         /* tests the filter passed on to the f64 test (RT_FLAG_DIAG_STATS, else 0) */
    uint64_t exact_roots;        /* f64 tests that reached the sqrt, sphere.rs:26   (RT_FLAG_DIAG_STATS, else 0) */
    float    kernel_ms;          /* render kernel, HIP events on its stream         */
    int32_t  n_spheres;
    int32_t  grid_blocks, block_threads;
    int32_t  scan_mode;          /* sphere-scan filter that ran: 0 none (RT_FLAG_NO_FILTER), 5 tube filter (shipped),
                                    1 VALU cross-check (RTIOW_SCAN_MODE=1); 2-4 only in RTIOW_CROSSCHECK_MODES builds */
    int32_t  kernel_variant;     /* which instantiation of the kernel ran, as bits: 1 the scan_mode-5 kernel for scenes whose
                                    tile grid has <= 64 cells (else the general one, and every other scan mode); 2 the
                                    RT_FLAG_UNIFORM53 instantiation; 4 work blocks of 1 024 pixel-samples instead of 256 (launches of
                                    >= 2 x 10^8 pixel-samples at >= 147 samples per pixel; >= 69 on the small-grid kernel) */
    uint64_t live_per_bounce[64]; /* rays traced at bounce index k (0 = camera ray; indices >= 63 share the
                                    last slot); sums to rays_traced (RT_FLAG_DIAG_STATS, else 0) */
    uint64_t direct_samples;     /* samples added to the frame buffer one by one instead of through their block's
                                    sums: all of them in a launch of < 5 spp (< 9 with RT_FLAG_NO_FILTER / RT_FLAG_DIAG_STATS), else only the last samples of blocks
                                    that a long path held open for too long (more than ~30 bounces at >= 37 spp, ~15 on the small-grid
                                    kernel; a few bounces on the 64-sample work blocks of launches with < 13 spp) */
} rt_stats;

/* ---- diagnostic knobs (environment; none changes a result) -------------------
 * Read by the library, for tests and measurements only -- every one of them selects another way of computing the
 * SAME frame (the parity tests run the knobs against the oracle):
 *   RTIOW_SCAN_MODE=1|5            rt_create: the sphere-scan filter (5 tube filter on the matrix pipe, shipped; 1 VALU cross-check)
 *   RTIOW_NO_GRID=1, RTIOW_GRID_DIM=G   rt_upload_scene: no tile grid / G x G cells instead of the cost model's choice
 *   RTIOW_BLOCKS_PER_CU=k          rt_create: workgroups per CU of the persistent grid (default: the occupancy query)
 *   RTIOW_RING_MIN_SPP=n           rt_create: no per-block pixel sums in LDS below n samples per pixel (default: wherever a block's pixels fit the sums' slots,
 *                                  i.e. from 5 samples per pixel on; 9 with RT_FLAG_NO_FILTER / RT_FLAG_DIAG_STATS)
 *   RTIOW_LARGE_BLOCK_MIN_ITEMS=n  per launch: work blocks of 1 024 pixel-samples instead of 256 from n pixel-samples per launch on
 *                                  (default 2 x 10^8; also needs >= 147 samples per pixel, 69 on the small-grid kernel; rt_stats.kernel_variant bit 2 says which ran) */

/* ---- lifetime -------------------------------------------------------------- */

/* Opens HIP device `device_id`.  Fails with RT_ERR_NO_DEVICE when there is no
 * HIP device or it is not gfx950 -- there is no fallback path. */
int rt_create(int32_t device_id, rt_context **out);
int rt_destroy(rt_context *ctx);

/* ---- scene: stands in for `&world` captured at main.rs:135 ----------------- */
/* The reference's list is a Vec<Box<dyn Hit>> of any length (shapes/mod.rs:52).  Here n <= RT_MAX_SPHERES: the
 * kernel numbers the columns of its filter table in 26 bits.  (Rounds 1-3 stopped at 65 535: 16-bit candidate numbers.)
 * A performance cliff sits far below the limit: the tile grid has at most 63 x 63 cells of 32 columns (+ 48 tiles every ray
 * scans), i.e. ~128 K columns; a scene that needs more -- upwards of ~130 000 filtered spheres -- gets NO grid, every wave
 * then scans all n/32 tiles per bounce (O(n) per ray, still exact: tests/test_limits.py renders 70 227 spheres both ways).
 * Coordinates and radii must be finite and below 1e15 in magnitude, radii non-zero, kinds RT_LAMBERTIAN / RT_METAL /
 * RT_DIALECTRIC. */
#define RT_MAX_SPHERES (1 << 24)
int rt_upload_scene(rt_context *ctx, const rt_sphere *spheres, int32_t n);

/* Rows owned by (shard_index, shard_count, tile_rows) of an image `height`
 * rows tall; and the image row j of compact row r. */
int rt_shard_rows(const rt_params *p, int32_t *out_rows);
int rt_shard_row_index(const rt_params *p, int32_t compact_row, int32_t *out_j);

/* ---- the hot path: main.rs:122-139 up to (not including) to_rgba ----------- */

/* The exact sums and their range (contract C5).  The reference adds every sample's colour into an f64 per pixel
 * (main.rs:127,135), unbounded.  Here one channel of one sample enters the pixel's sum as
 *     q = floor(min(x, RT_SAMPLE_CLAMP) * 2^32)   (0 for a NaN or a negative x)
 * and the sums are u64: exact, associative, identical however the samples are spread over lanes, launches or GPUs.
 * With q <= 2^48 a sum CANNOT wrap while a pixel has received FEWER THAN 65 536 samples (at most 65 535: all launches
 * that accumulate into the same buffer together; 65 536 saturated samples would sum to exactly 2^64), whatever the scene;
 * and within that limit the clamp cannot change a byte of
 * Color::to_rgba: a clamped sample alone puts the pixel's mean at >= 1, i.e. at byte 255, where the reference's
 * unbounded sum puts it too.  From 65 536 samples per pixel on the sums stay exact while the pixel's mean radiance is
 * below 2^32 / samples; a scene whose albedos are <= 1 (every scene of the reference) has x <= 1 and no limit below
 * 2^32 samples.  (Rounds 1-3 clamped at 2^30: four saturated samples wrapped a sum.) */
#define RT_SAMPLE_CLAMP 65536.0

/* Host-buffer form.  out_sum: [rows][width][3] f32 radiance SUMS over the spp
 * samples (divide by spp for the mean), rows = rt_shard_rows().  out_fix
 * (optional, may be NULL): the exact sums, u64 with quantum 2^-32.
 * Synchronous. */
int rt_render(rt_context *ctx, const rt_camera *cam, const rt_params *p,
              float *out_sum, uint64_t *out_fix, rt_stats *stats);

/* Device-buffer form, asynchronous on `stream` (a hipStream_t, or NULL for the
 * default stream).  d_fix: device pointer to [rows][width][3] u64.
 * Progressive passes (main.rs:130-137 split over launches with sample_begin and RT_FLAG_ACCUMULATE): the sums are exact
 * integers added with atomics, so passes may OVERLAP -- issue pass k + 1 on another stream than pass k (and say so: RT_FLAG_OVERLAPPED) and it fills the tail
 * of pass k (the last paths of a launch leave most of the chip idle: 5 % of a 100-spp launch at 1200x675).  A context holds
 * the per-launch state of two launches; a third launch makes ITS stream wait for the one before the previous (no host wait).
 * The caller orders what must be ordered: the buffer is zeroed (by a launch without RT_FLAG_ACCUMULATE, or by the caller)
 * before any other stream adds to it, and it is read after every stream that adds to it has been waited for.
 * rt_last_stats reports on the latest launch only. */
int rt_render_device(rt_context *ctx, const rt_camera *cam, const rt_params *p,
                     void *d_fix, void *stream);

/* Exact sums -> f32 sums, on the device (count = rows*width*3 values). */
int rt_fix_to_f32_device(rt_context *ctx, const void *d_fix, int64_t count,
                         void *d_out_f32, void *stream);

/* Waits for the context's last launch and returns its counters and time. */
int rt_last_stats(rt_context *ctx, rt_stats *stats);

/* ---- Color::to_rgba (vec3.rs:403-421) + the row flip (main.rs:141-145) ----- */

/* d_fix: device [rows][width][3] exact sums; d_rgba: device [rows][width][4]
 * u8.  spp = total samples in the sums.  Each channel is (f64)q * 2^-32 put
 * through vec3.rs:403-421 in f64.  flip != 0 writes row r at rows-1-r, which
 * for a whole image is the top-to-bottom order main.rs:141-145 produces. */
int rt_resolve_rgba8_device(rt_context *ctx, const void *d_fix, int32_t width, int32_t rows,
                            int64_t spp, int32_t flip, void *d_rgba, void *stream);
/* Host-buffer form (copies in, resolves on the device, copies out). */
int rt_resolve_rgba8(rt_context *ctx, const uint64_t *fix, int32_t width, int32_t rows,
                     int64_t spp, int32_t flip, uint8_t *out_rgba);

/* ---- the whole of main.rs:122-145 in ONE call ------------------------------- */

/* Renders (main.rs:122-136), keeps the exact sums ON THE DEVICE, puts them through Color::to_rgba with
 * p->spp samples (main.rs:137, vec3.rs:403-421) and the row flip (main.rs:141-145, flip != 0), and copies the
 * bytes out: out_rgba is [rows][width][4] u8, rows = rt_shard_rows() -- exactly the Vec<u8> that
 * ImageBuffer::from_vec takes at main.rs:147.  4 bytes per pixel cross PCIe, once (rt_render + rt_resolve_rgba8
 * move the 24-byte sums out and back in first).  Same bytes as that pair of calls.  Starts from zero
 * (RT_FLAG_ACCUMULATE is ignored, as in rt_render); synchronous.  stats may be NULL. */
int rt_render_rgba8(rt_context *ctx, const rt_camera *cam, const rt_params *p, int32_t flip,
                    uint8_t *out_rgba, rt_stats *stats);

/* ---- misc ------------------------------------------------------------------ */
const char *rt_last_error(void);
const char *rt_backend_name(void);     /* "hip-gfx950" */
int32_t rt_abi_version(void);
/* sha256 (first 16 hex digits) over the product kernel sources (csrc/rt_*.hpp, rt_api.hip) this library was BUILT from, as the build recorded it
 * ("unknown" for a build that did not pass it): bench.py labels its line with it, so that a stale .so shows */
const char *rt_build_source_sha(void);
/* Known-answer test hooks, computed ON THE DEVICE:
 * one Philox4x32-10 block; and elementwise a[i]/b[i] and sqrt(a[i]) in f64 (the
 * two operations whose correct rounding the bit-exact contract leans on). */
int rt_f64_div_sqrt_device(rt_context *ctx, const double *a, const double *b, int32_t n,
                           double *out_div, double *out_sqrt);
/* ... and the kernel's quantisation of one radiance channel to the exact 2^-32 grid (contract C5, DESIGN.md
 * section 4): out[i] = floor(min(x[i], RT_SAMPLE_CLAMP) * 2^32) for x[i] >= 0, and 0 for negatives and NaN. */
int rt_quantize_device(rt_context *ctx, const double *x, int32_t n, uint64_t *out);
/* ... and the kernel's rejection tests and word -> draw rules on n triples (wx, wy, wz) of Philox words: u01(w) = w * 2^-32 (draws
 * from [0,1)), u11(w) = (int32_t)w * 2^-31 (draws from (-1..1) and (-1..=1): the word as a two's-complement integer).  out_accept[k]
 * bit 0 = random_in_unit_sphere's `length_squared() < 1.0` (vec3.rs:37-45) for (u11(wx), u11(wy), u11(wz)), bit 1 =
 * random_in_unit_disk's (vec3.rs:59-68) for (u11(wx), u11(wy)), as the retry loops decide them (on the integers behind the draws,
 * with the f64 expression where its roundings could decide: DESIGN.md section 3); out_uniforms[k] = (u01(wx), u11(wx), u11(wy), u11(wz)). */
int rt_unit_accept_device(rt_context *ctx, const uint32_t *words, int32_t n, uint32_t *out_accept, double *out_uniforms);
/* Known-answer hooks of the EARLIER matrix-pipe forms of the filter (scan modes 2-4).
 * They exist only in a library built with -DRTIOW_CROSSCHECK_MODES (tools/librtiow_hip_xcheck.so, a test
 * artefact); the product library carries scan modes 0, 1 and 5 and does not export them. */
#ifdef RTIOW_CROSSCHECK_MODES
/* The two K = 4 products of the scan filter exactly as the render
 * kernel's matrix-pipe tiles evaluate them: r1, r2: [64][4] ray rows, s: [16][4] sphere
 * columns, out_hb, out_q: [64][16].  bf16x3 != 0 selects the three-piece bf16 form. */
int rt_filter_products_device(rt_context *ctx, const float *r1, const float *r2, const float *s,
                               int32_t bf16x3, float *out_hb, float *out_q);
/* One tile of the single-contraction ("lifted") form of the same filter (scan mode 4):
 * o, d: [64][3] f64 rays; spheres16: 16 spheres (one tile of columns, built exactly as
 * rt_upload_scene builds them); out_D: [64][16] the sums whose sign the kernel tests;
 * out_R: [64][11] the per-ray terms (the last entry: 1 if the ray is inside the analysed range);
 * out_C: [16][11] the per-sphere terms. */
int rt_filter_lifted_device(rt_context *ctx, const double *o, const double *d, const rt_sphere *spheres16,
                            float *out_D, float *out_R, float *out_C);
#endif /* RTIOW_CROSSCHECK_MODES */
/* One tile of the tube filter, the shipped scan mode: o, d: [64][3] f64 rays; spheres32: 32 spheres
 * (one tile of columns, built exactly as rt_upload_scene builds them, radius floor included);
 * out_h: [64][32][2] the two per-direction values H_k as the matrix pipe returns them, in units of HALF the
 * sphere's bound: the kernel keeps a (ray, sphere) pair iff |H_1| < 2 and |H_2| < 2 (it tests one bit of each);
 * out_bound[32]: the bound max(R, rho) each column was scaled with (sigma = 2 (1 - 2^-6) / bound, rounded down
 * to a bf16, is the value in K-slots 12..14 of the column); out_rows: [64][9] = (lambda u_1, lambda u_2, t_1, t_2,
 * 1 if the ray is inside the analysed range); out_rho: the radius floor chosen for these 32 spheres. */
int rt_filter_tube_device(rt_context *ctx, const double *o, const double *d, const rt_sphere *spheres32,
                          float *out_h, float *out_rows, float *out_bound, float *out_rho);
/* The host half of the same tile, no device needed: the 32 columns exactly as rt_upload_scene lays them
 * out.  out_words: [64][4] u32 = the B operand of lane l (column l&31, K-slots 8(l>>5)..+7, two bf16 per
 * word, low half first): slots 0..11 two bf16 pieces of sigma * centre per coordinate as (y1, y2, y1, y2),
 * slots 12..14 sigma, slot 15 zero (4.0 in a column no ray may keep); out_bound: [32]; out_rho: the radius floor. */
int rt_tube_tile_host(const rt_sphere *spheres32, uint32_t *out_words, float *out_bound, float *out_rho);
/* Where rt_upload_scene puts each sphere in the filter's table of columns, no device needed.  The table is made
 * of tiles of 32 columns: tiles [0, n_global) are scanned for every ray (spheres too large for a grid cell, and what
 * did not fit its cell's tile); tile n_global + iz * grid_dim + ix holds spheres whose centre lies in cell (ix, iz)
 * of a square grid over x and z.  out_dims = (grid_dim, n_global), grid_dim 0: no grid, columns in list order;
 * out_grid = (x0, z0, 1 / cell size, x1, z1, y lo, y hi, largest radius in a cell): every sphere of a cell tile has
 * its centre in that cell, its radius <= out_grid[7] and its extent in y within [out_grid[5], out_grid[6]];
 * out_slot_of[column] = the sphere's place in the list, -1 for padding.  Spheres on the always-exact list are in no
 * column.  Returns the number of columns (a multiple of 32, <= cap) or a negative RT_ERR_*. */
int rt_tile_layout_host(const rt_sphere *spheres, int32_t n, int32_t out_dims[2], float out_grid[8], int32_t *out_slot_of, int32_t cap);
int rt_philox_device(rt_context *ctx, const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

#ifdef __cplusplus
}
#endif
#endif
