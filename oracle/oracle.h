/*
 * oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the Druthyn/rtiow hot path (the pixel x sample Monte-Carlo
 * loop).  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * only as the checker / timed CPU baseline.  The product library
 * (librtiow_hip.so) never links, loads or calls anything declared here.
 *
 * Two oracles share this interface:
 *
 *   Oracle A  (oracle_a_f64.c)  literal f64 restatement of the reference:
 *             recursion, ordered list scan, no FMA contraction, the
 *             reference's operation order.  Semantic reference and the timed
 *             CPU baseline ("port").
 *   Oracle B  (oracle_b_f32.c)  the f32 "arithmetic contract" the HIP kernel
 *             implements: the same algorithm in f32 with
 *             a documented set of FMA fusions, iterative bounce loop, exact
 *             fixed-point accumulation.  Bit-level comparator for the GPU.
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
 * (src/main.rs:64,87,93-99).  Same 36-byte layout as rt_sphere in
 * include/rtiow_hip.h so tests hand the very same bytes to both sides. */
typedef struct {
    float center[3];
    float radius;
    int32_t kind;      /* 0 Lambertian, 1 Metal, 2 Dialectric (materials.rs) */
    float albedo[3];
    float param;       /* Metal: fuzz; Dialectric: ir */
} oracle_sphere;

/* camera.rs:4-13 minus `w` (unused by get_ray). */
typedef struct {
    double origin[3], lower_left_corner[3], horizontal[3], vertical[3];
    double u[3], v[3];
    double lens_radius;
} oracle_camera_f64;

typedef struct {
    float origin[3], lower_left_corner[3], horizontal[3], vertical[3];
    float u[3], v[3];
    float lens_radius;
} oracle_camera_f32;

typedef struct {
    int32_t width, height;
    int32_t spp;            /* samples rendered by this call            */
    int32_t sample_begin;   /* first sample index (additive passes)     */
    int32_t max_depth;      /* 50 in the reference (main.rs:28)         */
    int32_t row_begin, row_end, row_step; /* rows j (j=0 is the BOTTOM row) */
    uint64_t seed;
    double t_min;           /* 1e-4 in the reference (main.rs:44)       */
    int32_t nthreads;       /* <=0: all hardware threads                */
} oracle_params;

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

/* Camera::new (camera.rs:17-45) in f64; the f32 camera is its rounding. */
void oracle_camera_new(const double look_from[3], const double look_at[3], const double v_up[3],
                       double v_fov_deg, double aspect_ratio, double aperture, double focus_dist,
                       oracle_camera_f64 *out);
void oracle_camera_to_f32(const oracle_camera_f64 *in, oracle_camera_f32 *out);

/* Oracle A.  out_sum: [rows][W][3] f64 radiance SUMS over the spp samples
 * (rows = those selected by row_begin/end/step, ascending j). */
int oracle_a_render(const oracle_camera_f64 *cam, const oracle_sphere *spheres, int32_t n,
                    const oracle_params *p, double *out_sum, oracle_stats *stats);
void oracle_a_resolve_rgba8(const double *sum, int32_t width, int32_t rows, int64_t spp,
                            int32_t flip, uint8_t *out);

/* Oracle B.  out_fix: [rows][W][3] u64 fixed-point sums (quantum 2^-32);
 * out_sum (optional): the same converted to f32 sums. */
int oracle_b_render(const oracle_camera_f32 *cam, const oracle_sphere *spheres, int32_t n,
                    const oracle_params *p, uint64_t *out_fix, float *out_sum, oracle_stats *stats);
void oracle_b_fix_to_f32(const uint64_t *fix, int64_t count, float *out);
void oracle_b_resolve_rgba8(const float *sum, int32_t width, int32_t rows, int64_t spp,
                            int32_t flip, uint8_t *out);

/* Unit-level entry points for the analytic known-answer tests (f64 = A,
 * f32 = B).  hit: returns 1 and fills t,p,normal,front_face on a hit. */
int oracle_a_sphere_hit(const double c[3], double radius, const double o[3], const double d[3],
                        double t_min, double t_max, double *t, double p[3], double n[3], int *front);
int oracle_b_sphere_hit(const float c[3], float radius, const float o[3], const float d[3],
                        float t_min, float t_max, float *t, float p[3], float n[3], int *front);
/* world.hit over a list: returns index of the winner or -1 (tie rule test). */
int oracle_a_world_hit(const oracle_sphere *s, int32_t n, const double o[3], const double d[3],
                       double t_min, double *t);
int oracle_b_world_hit(const oracle_sphere *s, int32_t n, const float o[3], const float d[3],
                       float t_min, float *t);
void oracle_a_reflect(const double v[3], const double n[3], double out[3]);
void oracle_a_refract(const double uv[3], const double n[3], double ratio, double out[3]);
double oracle_a_reflectance(double cosine, double ref_idx);
void oracle_b_reflect(const float v[3], const float n[3], float out[3]);
void oracle_b_refract(const float uv[3], const float n[3], float ratio, float out[3]);
float oracle_b_reflectance(float cosine, float ref_idx);
/* scatter with an explicit uniform stream: u[] are U[0,1) draws consumed in
 * reference order.  Returns 1 scattered / 0 absorbed; *used = draws consumed. */
int oracle_a_scatter(const oracle_sphere *mat, const double d_in[3], const double p[3],
                     const double n[3], int front, const double *u, int nu, int *used,
                     double att[3], double d_out[3]);
int oracle_b_scatter(const oracle_sphere *mat, const float d_in[3], const float p[3],
                     const float n[3], int front, const float *u, int nu, int *used,
                     float att[3], float d_out[3]);
/* to_rgba of one colour (vec3.rs:403-421). */
void oracle_a_to_rgba(const double c[3], int64_t spp, uint8_t out[4]);
void oracle_b_to_rgba(const float c[3], int64_t spp, uint8_t out[4]);
/* quantisation of one f32 radiance value to the 2^-32 fixed-point grid. */
uint64_t oracle_b_quantize(float x);

int oracle_hardware_threads(void);

#ifdef __cplusplus
}
#endif
#endif
