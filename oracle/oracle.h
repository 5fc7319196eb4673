/*
 * oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the Druthyn/rtiow hot path (the pixel x sample Monte-Carlo
 * loop).  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * only as the checker / timed CPU baseline.  The product library
 * (librtiow_hip.so) never links, loads or calls anything declared here.
 *
 * Two oracles share this interface (both in oracle_f64.c, sharing every
 * function that restates the reference's geometry and materials):
 *
 *   Oracle A  literal f64 restatement of the reference: recursion
 *             (main.rs:38-57), ordered list scan, the reference's operation
 *             order, no FMA contraction, f64 sequential pixel sum.  Semantic
 *             reference and the timed CPU baseline ("port").
 *   Oracle B  the ARITHMETIC CONTRACT of the HIP kernel:
 *             Oracle A with exactly two changes -- C3: the recursion is an
 *             iterative bounce loop whose attenuations multiply left to
 *             right; C5: a pixel's sum is exact fixed point (each sample
 *             truncated to 2^-32, summed in u64).  Everything else, down to
 *             every rounding, is Oracle A's code.  Bit-level comparator for
 *             the GPU; it differs from A by a few 2^-32 quanta per pixel.
 *
 * Pinning: the reference has no tests and cannot be
 * built here (no Rust toolchain).  Pins: Philox4x32-10 Random123 KATs, the
 * exact sky pixels of the reference's committed render rtiow_part1_final.png
 * (tests/golden/ref_png_sky_rows.json) and build-authored analytic KATs.
 * The reference's RNG (rand 0.8.5 ThreadRng, OS seeded) is unreproducible:
 * parity is unpinned at the rand boundary; only the distributions are kept.
 */
#ifndef RTIOW_ORACLE_H
#define RTIOW_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One sphere of the flat scene, list order = reference push order
 * (src/main.rs:64,87,93-99).  Same 72-byte layout as rt_sphere in
 * include/rtiow_hip.h so tests hand the very same bytes to both sides. */
typedef struct {
    double center[3];
    double radius;
    double albedo[3];
    double param;      /* Metal: fuzz; Dialectric: ir */
    int32_t kind;      /* 0 Lambertian, 1 Metal, 2 Dialectric (materials.rs) */
    int32_t reserved;
} oracle_sphere;

/* camera.rs:4-13 minus `w` (unused by get_ray). */
typedef struct {
    double origin[3], lower_left_corner[3], horizontal[3], vertical[3];
    double u[3], v[3];
    double lens_radius;
} oracle_camera;

typedef struct {
    int32_t width, height;
    int32_t spp;            /* samples rendered by this call            */
    int32_t sample_begin;   /* first sample index (additive passes)     */
    int32_t max_depth;      /* 50 in the reference (main.rs:28)         */
    int32_t row_begin, row_end, row_step; /* rows j (j=0 is the BOTTOM row) */
    uint64_t seed;
    double t_min;           /* 1e-4 in the reference (main.rs:44)       */
    int32_t nthreads;       /* <=0: all hardware threads                */
    uint32_t flags;         /* ORACLE_FLAG_*                            */
} oracle_params;
/* Every uniform from TWO consecutive Philox words, u = ((w0 << 32 | w1) >> 11) * 2^-53 (the 53 random bits of rand 0.8.5's
 * gen::<f64>(), main.rs:131-132, vec3.rs:31-33,63, materials.rs:96) instead of one word's 32 bits; same draw order, same
 * runs of consecutive words (oracle_common.h).  Mirrors RT_FLAG_UNIFORM53 of include/rtiow_hip.h. */
#define ORACLE_FLAG_UNIFORM53 0x8u

typedef struct {
    uint64_t samples;
    uint64_t rays_traced;       /* world.hit() calls                    */
    uint64_t depth_hist[64];    /* paths that ended after k scatters    */
    uint64_t end_sky, end_absorb, end_depth;
    double seconds;
    int32_t threads_used;
} oracle_stats;

/* Philox4x32-10 (Random123 spec), exported for the KAT test. */
void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* Camera::new (camera.rs:17-45) in f64. */
void oracle_camera_new(const double look_from[3], const double look_at[3], const double v_up[3],
                       double v_fov_deg, double aspect_ratio, double aperture, double focus_dist,
                       oracle_camera *out);

/* Oracle A.  out_sum: [rows][W][3] f64 radiance SUMS over the spp samples
 * (rows = those selected by row_begin/end/step, ascending j). */
int oracle_a_render(const oracle_camera *cam, const oracle_sphere *spheres, int32_t n,
                    const oracle_params *p, double *out_sum, oracle_stats *stats);
/* Color::to_rgba (vec3.rs:403-421) + row flip (main.rs:141-145) from f64 sums. */
void oracle_a_resolve_rgba8(const double *sum, int32_t width, int32_t rows, int64_t spp,
                            int32_t flip, uint8_t *out);

/* Oracle B.  out_fix: [rows][W][3] u64 fixed-point sums (quantum 2^-32);
 * out_sum (optional): the same as f32 sums, (f32)((f64)q * 2^-32). */
int oracle_b_render(const oracle_camera *cam, const oracle_sphere *spheres, int32_t n,
                    const oracle_params *p, uint64_t *out_fix, float *out_sum, oracle_stats *stats);
void oracle_b_fix_to_f32(const uint64_t *fix, int64_t count, float *out);
/* to_rgba in f64 of the exact sums: c = (f64)q * 2^-32, then vec3.rs:403-421. */
void oracle_b_resolve_rgba8(const uint64_t *fix, int32_t width, int32_t rows, int64_t spp,
                            int32_t flip, uint8_t *out);
/* C5: truncation of one radiance value to the 2^-32 grid. */
uint64_t oracle_b_quantize(double x);

/* Unit-level entry points for the analytic known-answer tests.
 * hit: returns 1 and fills t,p,normal,front_face on a hit. */
int oracle_sphere_hit(const double c[3], double radius, const double o[3], const double d[3],
                      double t_min, double t_max, double *t, double p[3], double n[3], int *front);
/* world.hit over a list: returns index of the winner or -1 (tie rule test). */
int oracle_world_hit(const oracle_sphere *s, int32_t n, const double o[3], const double d[3],
                     double t_min, double *t);
void oracle_reflect(const double v[3], const double n[3], double out[3]);
void oracle_refract(const double uv[3], const double n[3], double ratio, double out[3]);
double oracle_reflectance(double cosine, double ref_idx);
/* scatter with an explicit uniform stream: u[] are U[0,1) draws consumed in
 * reference order.  Returns 1 scattered / 0 absorbed; *used = draws consumed. */
int oracle_scatter(const oracle_sphere *mat, const double d_in[3], const double p[3],
                   const double n[3], int front, const double *u, int nu, int *used,
                   double att[3], double d_out[3]);
/* to_rgba of one colour (vec3.rs:403-421). */
void oracle_to_rgba(const double c[3], int64_t spp, uint8_t out[4]);
/* one camera ray for given (s,t) and lens sample (camera.rs:47-54). */
void oracle_get_ray(const oracle_camera *cam, double s, double t, double lens_x, double lens_y,
                    double orig[3], double dir[3]);

/* first `count` draws of the stream of (pixel, sample) as one run: from [0,1) (u = w * 2^-32; 53-bit with ORACLE_FLAG_UNIFORM53),
 * or with ORACLE_FLAG_SYMMETRIC_DRAWS (this hook only) from the symmetric ranges (x = (int32_t)w * 2^-31; 2u - 1 with 53 bits) */
#define ORACLE_FLAG_SYMMETRIC_DRAWS 0x100u
void oracle_uniforms(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t flags, int32_t count, double *out);

int oracle_hardware_threads(void);

#ifdef __cplusplus
}
#endif
#endif
