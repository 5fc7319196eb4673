/*
 * oracle_f64.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Oracle A: LITERAL f64 restatement of the reference hot path.  Recursion,
 * the ordered linear scan, the reference's operation order; compiled with
 * -ffp-contract=off so no multiply-add is fused (Rust never fuses).  Every
 * function cites the reference lines it follows (paths under /root/reference).
 * The only thing that is not the reference's is the source of uniforms
 * (oracle_common.h): same distributions, same draw order.
 *
 * Oracle B: the arithmetic contract of the HIP kernel = Oracle A with
 *   C3  ray_color's recursion (main.rs:38-57) as an iterative bounce loop:
 *       attenuations multiply into a running throughput left to right and
 *       the sky colour is multiplied by the throughput at the end;
 *   C5  an exact pixel sum: each sample's radiance channel x is truncated to
 *       the 2^-32 grid, q = trunc(clamp(x, 0, 2^16) * 2^32) (NaN -> 0), and
 *       summed in a u64 (associative: any order or sharding of the samples
 *       gives the same bits).
 * B calls the very same sphere_hit / world_hit / scatter / get_ray functions
 * as A; only ray_color and the accumulation differ.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle_common.h"

typedef struct { double x, y, z; } vec3;

/* ---- vec3.rs ------------------------------------------------------------ */
static inline vec3 v3(double x, double y, double z) { vec3 r = { x, y, z }; return r; }
static inline vec3 add(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }       /* :137-147 */
static inline vec3 sub(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }       /* :243-253 */
static inline vec3 muls(vec3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }          /* :330-356 */
static inline vec3 mulv(vec3 a, vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }      /* :358-367 */
static inline vec3 divs(vec3 a, double s) { return muls(a, 1.0 / s); }                       /* :371-375 */
static inline double dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }       /* :95-97  */
static inline double length_squared(vec3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }    /* :87-89 powi(2) */
static inline double length(vec3 a) { return sqrt(length_squared(a)); }                      /* :83-85  */
static inline vec3 unit_vector(vec3 a) { return divs(a, length(a)); }                        /* :107-109 */
static inline int is_near_zero(vec3 a)
{   /* :111-114 */
    const double s = 1e-8;
    return fabs(a.x) < s && fabs(a.y) < s && fabs(a.z) < s;
}
static inline vec3 reflect(vec3 v, vec3 n)
{   /* :116-118  self - 2.0*self.dot(n) * *n  ==  v - ((2.0*dot) * n) */
    return sub(v, muls(n, 2.0 * dot(v, n)));
}
static inline double min_1(double x) { return (x < 1.0) ? x : 1.0; }  /* 1.0_f64.min(x): NaN -> 1.0 */
static inline vec3 refract(vec3 uv, vec3 n, double etai_over_etat)
{   /* :120-125 */
    double cos_theta = min_1(-dot(uv, n));
    vec3 r_out_perp = muls(add(uv, muls(n, cos_theta)), etai_over_etat);
    vec3 r_out_parallel = muls(n, -sqrt(fabs(1.0 - length_squared(r_out_perp))));
    return add(r_out_perp, r_out_parallel);
}

/* random_in_range(-1,1) components (:26-35) and random_in_unit_sphere (:37-45):
 * one run of (x,y,z) tries, three consecutive words each; gen_range(-1.0..=1.0) -> rng_take_sym. */
static vec3 random_in_unit_sphere(oracle_rng *rng)
{
    for (;;) {
        double u[3];
        rng_take_sym(rng, 3, u);
        vec3 p = v3(u[0], u[1], u[2]);
        if (length_squared(p) < 1.0) { rng_end_run(rng); return p; }
    }
}
static vec3 random_unit_vector(oracle_rng *rng) { return unit_vector(random_in_unit_sphere(rng)); } /* :47-49 */

/* ---- ray.rs:15-17 -------------------------------------------------------- */
typedef struct { vec3 orig, dir; } ray;
static inline vec3 ray_at(const ray *r, double t) { return add(r->orig, muls(r->dir, t)); }

/* ---- shapes/mod.rs:10-30 -------------------------------------------------- */
typedef struct { vec3 p, normal; int mat; double t; int front_face; } hit_record;

static inline void hit_record_new(hit_record *rec, vec3 p, double t, const ray *r, vec3 outward_normal, int mat)
{   /* mod.rs:20-30 */
    int front_face = dot(r->dir, outward_normal) < 0.0;
    rec->p = p; rec->t = t; rec->mat = mat; rec->front_face = front_face;
    rec->normal = front_face ? outward_normal : sub(v3(0.0, 0.0, 0.0), outward_normal);
}

/* ---- shapes/sphere.rs:15-41 ---------------------------------------------- */
static inline int sphere_hit(vec3 center, double radius, int mat, const ray *r,
                             double t_min, double t_max, hit_record *rec)
{
    vec3 oc = sub(r->orig, center);
    double a = length_squared(r->dir);
    double half_b = dot(oc, r->dir);
    double c = length_squared(oc) - radius * radius;
    double discriminant = half_b * half_b - a * c;
    if (discriminant < 0.0) return 0;
    double sqrtd = sqrt(discriminant);
    double root = (-half_b - sqrtd) / a;
    if (root < t_min || t_max < root) {
        root = (-half_b + sqrtd) / a;
        if (root < t_min || t_max < root) return 0;
    }
    vec3 p = ray_at(r, root);
    vec3 outward_normal = divs(sub(p, center), radius);
    hit_record_new(rec, p, root, r, outward_normal, mat);
    return 1;
}

typedef struct { vec3 center; double radius; int kind; vec3 albedo; double param; } sphere64;

/* ---- shapes/mod.rs:54-70 -------------------------------------------------- */
static int world_hit(const sphere64 *w, int n, const ray *r, double t_min, double t_max, hit_record *out)
{
    int found = 0;
    double closest_so_far = t_max;
    hit_record rec;
    for (int i = 0; i < n; ++i) {
        if (sphere_hit(w[i].center, w[i].radius, i, r, t_min, closest_so_far, &rec)) {
            closest_so_far = rec.t;
            *out = rec;
            found = 1;
        }
    }
    return found;
}

/* ---- materials.rs --------------------------------------------------------- */
static double reflectence(double cosine, double ref_idx)
{   /* :78-82; powi(2) = x*x, powi(5) = ((x*x)*(x*x))*x */
    double r0 = (1.0 - ref_idx) / (1.0 + ref_idx);
    r0 = r0 * r0;
    double x = 1.0 - cosine;
    double x2 = x * x;
    double x5 = (x2 * x2) * x;
    return r0 + (1.0 - r0) * x5;
}

static int scatter(const sphere64 *m, const ray *r_in, const hit_record *rec, oracle_rng *rng,
                   vec3 *att, ray *scattered)
{
    if (m->kind == 0) {                       /* Lambertian, :21-31 */
        vec3 scatter_direction = add(rec->normal, random_unit_vector(rng));
        if (is_near_zero(scatter_direction)) scatter_direction = rec->normal;
        scattered->orig = rec->p; scattered->dir = scatter_direction;
        *att = m->albedo;
        return 1;
    } else if (m->kind == 1) {                /* Metal, :48-62 */
        vec3 reflected = unit_vector(reflect(r_in->dir, rec->normal));
        scattered->orig = rec->p;
        scattered->dir = add(reflected, muls(random_in_unit_sphere(rng), m->param));
        if (dot(scattered->dir, rec->normal) <= 0.0) return 0;
        *att = m->albedo;
        return 1;
    } else {                                  /* Dialectric, :76-105 */
        double refraction_ratio = rec->front_face ? 1.0 / m->param : m->param;
        vec3 unit_direction = unit_vector(r_in->dir);
        double cos_theta = min_1(-dot(unit_direction, rec->normal));
        double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
        int can_refract = refraction_ratio * sin_theta <= 1.0;
        vec3 direction;
        int do_refract = 0;
        if (can_refract) {                    /* && short-circuit: draw only if can_refract */
            double u;
            double refl = reflectence(cos_theta, refraction_ratio);
            rng_take(rng, 1, &u);
            rng_end_run(rng);
            do_refract = refl <= u;
        }
        if (do_refract) direction = refract(unit_direction, rec->normal, refraction_ratio);
        else direction = reflect(unit_direction, rec->normal);
        scattered->orig = rec->p; scattered->dir = direction;
        *att = v3(1.0, 1.0, 1.0);
        return 1;
    }
}

/* ---- main.rs:38-57 -------------------------------------------------------- */
typedef struct {
    const sphere64 *world; int n; double t_min; int max_depth;
    uint64_t rays; uint64_t depth_hist[64]; uint64_t end_sky, end_absorb, end_depth;
} trace_ctx;

static vec3 ray_color(const ray *r, trace_ctx *cx, oracle_rng *rng, int depth)
{
    int k = cx->max_depth - depth;            /* scatters so far (stats only) */
    if (depth <= 0) {
        cx->end_depth++; cx->depth_hist[k < 63 ? k : 63]++;
        return v3(0.0, 0.0, 0.0);
    }
    hit_record rec;
    cx->rays++;
    if (world_hit(cx->world, cx->n, r, cx->t_min, INFINITY, &rec)) {
        vec3 att; ray scat;
        if (scatter(&cx->world[rec.mat], r, &rec, rng, &att, &scat))
            return mulv(att, ray_color(&scat, cx, rng, depth - 1));
        cx->end_absorb++; cx->depth_hist[k < 63 ? k : 63]++;
        return v3(0.0, 0.0, 0.0);
    }
    cx->end_sky++; cx->depth_hist[k < 63 ? k : 63]++;
    vec3 unit_direction = unit_vector(r->dir);
    double t = 0.5 * (unit_direction.y + 1.0);
    return add(muls(v3(1.0, 1.0, 1.0), 1.0 - t), muls(v3(0.5, 0.7, 1.0), t));
}

/* ---- camera.rs:47-54 ------------------------------------------------------ */
static ray get_ray(const oracle_camera *c, double s, double t, double lens_x, double lens_y)
{
    vec3 cu = v3(c->u[0], c->u[1], c->u[2]), cv = v3(c->v[0], c->v[1], c->v[2]);
    vec3 origin = v3(c->origin[0], c->origin[1], c->origin[2]);
    vec3 llc = v3(c->lower_left_corner[0], c->lower_left_corner[1], c->lower_left_corner[2]);
    vec3 hor = v3(c->horizontal[0], c->horizontal[1], c->horizontal[2]);
    vec3 ver = v3(c->vertical[0], c->vertical[1], c->vertical[2]);
    vec3 rd = muls(v3(lens_x, lens_y, 0.0), c->lens_radius);
    vec3 offset = add(muls(cu, rd.x), muls(cv, rd.y));
    ray r;
    r.orig = add(origin, offset);
    r.dir = sub(sub(add(add(llc, muls(hor, s)), muls(ver, t)), origin), offset);
    return r;
}

/* ---- main.rs:131-134: the camera ray of one pixel sample -------------------- */
static ray sample_ray(const oracle_camera *cam, const oracle_params *p, int i, int j, int s, oracle_rng *rng)
{
    rng_init(rng, p->seed, (uint32_t)j * (uint32_t)p->width + (uint32_t)i, (uint32_t)s);
    rng->u53 = (p->flags & ORACLE_FLAG_UNIFORM53) != 0;
    double e[4];
    rng_take(rng, 2, e);                                       /* B_0: one run with the lens tries that follow */
    rng_take_sym(rng, 2, e + 2);
    double u = ((double)i + e[0]) / (double)(p->width - 1);    /* main.rs:131 */
    double v = ((double)j + e[1]) / (double)(p->height - 1);   /* main.rs:132 */
    /* random_in_unit_disk, vec3.rs:59-68: gen_range(-1.0..1.0) -> rng_take_sym */
    double lx = e[2], ly = e[3];
    while (!(length_squared(v3(lx, ly, 0.0)) < 1.0)) {
        rng_take_sym(rng, 2, e);
        lx = e[0]; ly = e[1];
    }
    rng_end_run(rng);
    return get_ray(cam, u, v, lx, ly);
}

/* main.rs:135, Oracle A: recursive ray_color */
static vec3 sample_pixel(const oracle_camera *cam, trace_ctx *cx, const oracle_params *p,
                         int i, int j, int s)
{
    oracle_rng rng;
    ray r = sample_ray(cam, p, i, j, s, &rng);
    return ray_color(&r, cx, &rng, p->max_depth);
}

/* ---- Oracle B: contract C3 (iterative ray_color) ---------------------------- */
static vec3 ray_color_iter(ray r, trace_ctx *cx, oracle_rng *rng)
{
    vec3 thr = v3(1.0, 1.0, 1.0);
    int depth = cx->max_depth;
    for (;;) {
        int k = cx->max_depth - depth;
        if (depth <= 0) {                                   /* main.rs:40-42 */
            cx->end_depth++; cx->depth_hist[k < 63 ? k : 63]++;
            return v3(0.0, 0.0, 0.0);
        }
        hit_record rec;
        cx->rays++;
        if (!world_hit(cx->world, cx->n, &r, cx->t_min, INFINITY, &rec)) {
            cx->end_sky++; cx->depth_hist[k < 63 ? k : 63]++;
            vec3 unit_direction = unit_vector(r.dir);       /* main.rs:54-56 */
            double t = 0.5 * (unit_direction.y + 1.0);
            vec3 sky = add(muls(v3(1.0, 1.0, 1.0), 1.0 - t), muls(v3(0.5, 0.7, 1.0), t));
            return mulv(thr, sky);
        }
        vec3 att; ray scat;
        if (!scatter(&cx->world[rec.mat], &r, &rec, rng, &att, &scat)) {
            cx->end_absorb++; cx->depth_hist[k < 63 ? k : 63]++;
            return v3(0.0, 0.0, 0.0);                       /* main.rs:51 */
        }
        thr = mulv(thr, att);
        r = scat;
        depth -= 1;
    }
}

uint64_t oracle_b_quantize(double x)
{   /* C5 */
    if (!(x >= 0.0)) return 0;                  /* NaN and negatives */
    if (x > 65536.0) x = 65536.0;               /* 2^16: include/rtiow_hip.h RT_SAMPLE_CLAMP */
    return (uint64_t)(x * 4294967296.0);        /* exact scaling, truncation */
}

typedef struct {
    const oracle_camera *cam; const sphere64 *world; int n; const oracle_params *p;
    double *out; trace_ctx *ctxs;
} job_a;

/* Work unit of the thread pool: kSegCols consecutive pixels of one row (pixels are independent,
 * main.rs:122-139; whole rows would leave a one-row render on one thread). */
enum { kSegCols = 64 };
static int row_segments(const oracle_params *p) { return (p->width + kSegCols - 1) / kSegCols; }

static void row_a(void *arg, int unit, int worker)
{
    job_a *jb = (job_a *)arg;
    const oracle_params *p = jb->p;
    int step = p->row_step > 0 ? p->row_step : 1;
    int nseg = row_segments(p), slot = unit / nseg, seg = unit % nseg;
    int j = p->row_begin + slot * step;
    int i_end = (seg + 1) * kSegCols < p->width ? (seg + 1) * kSegCols : p->width;
    trace_ctx *cx = &jb->ctxs[worker];
    for (int i = seg * kSegCols; i < i_end; ++i) {
        vec3 pixel_color = v3(0.0, 0.0, 0.0);
        for (int s = p->sample_begin; s < p->sample_begin + p->spp; ++s)
            pixel_color = add(pixel_color, sample_pixel(jb->cam, cx, p, i, j, s));   /* main.rs:135 */
        double *o = jb->out + ((size_t)slot * p->width + i) * 3;
        o[0] = pixel_color.x; o[1] = pixel_color.y; o[2] = pixel_color.z;
    }
}

static sphere64 *promote_scene(const oracle_sphere *s, int n)
{
    sphere64 *w = (sphere64 *)malloc(sizeof(sphere64) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) {
        w[i].center = v3(s[i].center[0], s[i].center[1], s[i].center[2]);
        w[i].radius = s[i].radius; w[i].kind = s[i].kind;   /* f64 in, f64 kept: no rounding */
        w[i].albedo = v3(s[i].albedo[0], s[i].albedo[1], s[i].albedo[2]);
        w[i].param = s[i].param;
    }
    return w;
}

int oracle_a_render(const oracle_camera *cam, const oracle_sphere *spheres, int32_t n,
                    const oracle_params *p, double *out_sum, oracle_stats *stats)
{
    if (!cam || !p || !out_sum || n < 0 || p->width < 2 || p->height < 2 || p->spp < 0) return -1;
    int nrows = params_rows(p);
    int nthreads = p->nthreads > 0 ? p->nthreads : oracle_hardware_threads();
    if (nthreads > 256) nthreads = 256;
    sphere64 *world = promote_scene(spheres, n);
    trace_ctx *ctxs = (trace_ctx *)calloc((size_t)nthreads, sizeof(trace_ctx));
    for (int t = 0; t < nthreads; ++t) {
        ctxs[t].world = world; ctxs[t].n = n; ctxs[t].t_min = p->t_min; ctxs[t].max_depth = p->max_depth;
    }
    job_a jb = { cam, world, n, p, out_sum, ctxs };
    double t0 = oracle_now_seconds();
    int used = oracle_parallel_rows(nrows * row_segments(p), nthreads, row_a, &jb);
    double t1 = oracle_now_seconds();
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->samples = (uint64_t)nrows * (uint64_t)p->width * (uint64_t)p->spp;
        for (int t = 0; t < nthreads; ++t) {
            stats->rays_traced += ctxs[t].rays;
            stats->end_sky += ctxs[t].end_sky; stats->end_absorb += ctxs[t].end_absorb;
            stats->end_depth += ctxs[t].end_depth;
            for (int k = 0; k < 64; ++k) stats->depth_hist[k] += ctxs[t].depth_hist[k];
        }
        stats->seconds = t1 - t0; stats->threads_used = used;
    }
    free(ctxs); free(world);
    return used < 0 ? -2 : 0;
}

/* ---- Oracle B driver -------------------------------------------------------- */
typedef struct {
    const oracle_camera *cam; const sphere64 *world; int n; const oracle_params *p;
    uint64_t *out; trace_ctx *ctxs;
} job_b;

static void row_b(void *arg, int unit, int worker)
{
    job_b *jb = (job_b *)arg;
    const oracle_params *p = jb->p;
    int step = p->row_step > 0 ? p->row_step : 1;
    int nseg = row_segments(p), slot = unit / nseg, seg = unit % nseg;
    int j = p->row_begin + slot * step;
    int i_end = (seg + 1) * kSegCols < p->width ? (seg + 1) * kSegCols : p->width;
    trace_ctx *cx = &jb->ctxs[worker];
    for (int i = seg * kSegCols; i < i_end; ++i) {
        uint64_t acc[3] = { 0, 0, 0 };
        for (int s = p->sample_begin; s < p->sample_begin + p->spp; ++s) {
            oracle_rng rng;
            ray r = sample_ray(jb->cam, p, i, j, s, &rng);
            vec3 c = ray_color_iter(r, cx, &rng);
            acc[0] += oracle_b_quantize(c.x); acc[1] += oracle_b_quantize(c.y); acc[2] += oracle_b_quantize(c.z);
        }
        uint64_t *o = jb->out + ((size_t)slot * p->width + i) * 3;
        o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2];
    }
}

void oracle_b_fix_to_f32(const uint64_t *fix, int64_t count, float *out)
{   /* (f64)hi*2^32 + (f64)lo (one f64 rounding above 2^53), * 2^-32, to f32 */
    for (int64_t k = 0; k < count; ++k) {
        double d = (double)(uint32_t)(fix[k] >> 32) * 4294967296.0 + (double)(uint32_t)fix[k];
        out[k] = (float)(d * (1.0 / 4294967296.0));
    }
}

int oracle_b_render(const oracle_camera *cam, const oracle_sphere *spheres, int32_t n,
                    const oracle_params *p, uint64_t *out_fix, float *out_sum, oracle_stats *stats)
{
    if (!cam || !p || !out_fix || n < 0 || p->width < 2 || p->height < 2 || p->spp < 0) return -1;
    int nrows = params_rows(p);
    int nthreads = p->nthreads > 0 ? p->nthreads : oracle_hardware_threads();
    if (nthreads > 256) nthreads = 256;
    sphere64 *world = promote_scene(spheres, n);
    trace_ctx *ctxs = (trace_ctx *)calloc((size_t)nthreads, sizeof(trace_ctx));
    for (int t = 0; t < nthreads; ++t) {
        ctxs[t].world = world; ctxs[t].n = n; ctxs[t].t_min = p->t_min; ctxs[t].max_depth = p->max_depth;
    }
    job_b jb = { cam, world, n, p, out_fix, ctxs };
    double t0 = oracle_now_seconds();
    int used = oracle_parallel_rows(nrows * row_segments(p), nthreads, row_b, &jb);
    double t1 = oracle_now_seconds();
    if (out_sum) oracle_b_fix_to_f32(out_fix, (int64_t)nrows * p->width * 3, out_sum);
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->samples = (uint64_t)nrows * (uint64_t)p->width * (uint64_t)p->spp;
        for (int t = 0; t < nthreads; ++t) {
            stats->rays_traced += ctxs[t].rays;
            stats->end_sky += ctxs[t].end_sky; stats->end_absorb += ctxs[t].end_absorb;
            stats->end_depth += ctxs[t].end_depth;
            for (int k = 0; k < 64; ++k) stats->depth_hist[k] += ctxs[t].depth_hist[k];
        }
        stats->seconds = t1 - t0; stats->threads_used = used;
    }
    free(ctxs); free(world);
    return used < 0 ? -2 : 0;
}

/* ---- vec3.rs:403-421 + row flip main.rs:141-145 ---------------------------- */
static uint8_t as_u8(double x)
{   /* Rust `as u8`: saturating, NaN -> 0 */
    if (!(x == x)) return 0;
    if (x <= 0.0) return 0;
    if (x >= 255.0) return 255;
    return (uint8_t)x;
}
static double clamp_r(double x, double lo, double hi)
{   /* f64::clamp: NaN stays NaN */
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}
void oracle_to_rgba(const double c[3], int64_t spp, uint8_t out[4])
{
    double scale = 1.0 / (double)spp;
    for (int k = 0; k < 3; ++k) {
        double r = sqrt(scale * c[k]);
        out[k] = as_u8(256.0 * clamp_r(r, 0.0, 0.999));
    }
    out[3] = 255;
}
void oracle_a_resolve_rgba8(const double *sum, int32_t width, int32_t rows, int64_t spp,
                            int32_t flip, uint8_t *out)
{
    for (int r = 0; r < rows; ++r) {
        int dst = flip ? rows - 1 - r : r;
        for (int i = 0; i < width; ++i)
            oracle_to_rgba(sum + ((size_t)r * width + i) * 3, spp, out + ((size_t)dst * width + i) * 4);
    }
}

/* ---- unit-level exports ---------------------------------------------------- */
int oracle_sphere_hit(const double c[3], double radius, const double o[3], const double d[3],
                        double t_min, double t_max, double *t, double p[3], double n[3], int *front)
{
    ray r = { v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]) };
    hit_record rec;
    if (!sphere_hit(v3(c[0], c[1], c[2]), radius, 0, &r, t_min, t_max, &rec)) return 0;
    *t = rec.t; *front = rec.front_face;
    p[0] = rec.p.x; p[1] = rec.p.y; p[2] = rec.p.z;
    n[0] = rec.normal.x; n[1] = rec.normal.y; n[2] = rec.normal.z;
    return 1;
}
int oracle_world_hit(const oracle_sphere *s, int32_t n, const double o[3], const double d[3],
                       double t_min, double *t)
{
    sphere64 *w = promote_scene(s, n);
    ray r = { v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]) };
    hit_record rec;
    int idx = -1;
    if (world_hit(w, n, &r, t_min, INFINITY, &rec)) { idx = rec.mat; *t = rec.t; }
    free(w);
    return idx;
}
void oracle_reflect(const double v[3], const double n[3], double out[3])
{
    vec3 r = reflect(v3(v[0], v[1], v[2]), v3(n[0], n[1], n[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void oracle_refract(const double uv[3], const double n[3], double ratio, double out[3])
{
    vec3 r = refract(v3(uv[0], uv[1], uv[2]), v3(n[0], n[1], n[2]), ratio);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
double oracle_reflectance(double cosine, double ref_idx) { return reflectence(cosine, ref_idx); }

int oracle_scatter(const oracle_sphere *mat, const double d_in[3], const double p[3],
                     const double n[3], int front, const double *u, int nu, int *used,
                     double att[3], double d_out[3])
{
    sphere64 *m = promote_scene(mat, 1);
    oracle_rng rng;
    rng_init(&rng, 0, 0, 0);
    static const double none = 0.0;
    rng.explicit_u = u ? u : &none; rng.explicit_n = nu; rng.explicit_used = 0;
    ray r_in = { v3(0.0, 0.0, 0.0), v3(d_in[0], d_in[1], d_in[2]) };
    hit_record rec;
    rec.p = v3(p[0], p[1], p[2]); rec.normal = v3(n[0], n[1], n[2]);
    rec.front_face = front; rec.t = 0.0; rec.mat = 0;
    vec3 a = v3(0.0, 0.0, 0.0); ray sc = { rec.p, v3(0.0, 0.0, 0.0) };
    int ok = scatter(m, &r_in, &rec, &rng, &a, &sc);
    if (used) *used = rng.explicit_used;
    att[0] = a.x; att[1] = a.y; att[2] = a.z;
    d_out[0] = sc.dir.x; d_out[1] = sc.dir.y; d_out[2] = sc.dir.z;
    free(m);
    return ok;
}

void oracle_b_resolve_rgba8(const uint64_t *fix, int32_t width, int32_t rows, int64_t spp,
                            int32_t flip, uint8_t *out)
{
    for (int r = 0; r < rows; ++r) {
        int dst = flip ? rows - 1 - r : r;
        for (int i = 0; i < width; ++i) {
            const uint64_t *q = fix + ((size_t)r * width + i) * 3;
            double c[3];
            for (int k = 0; k < 3; ++k)
                c[k] = ((double)(uint32_t)(q[k] >> 32) * 4294967296.0 + (double)(uint32_t)q[k]) * (1.0 / 4294967296.0);
            oracle_to_rgba(c, spp, out + ((size_t)dst * width + i) * 4);
        }
    }
}

void oracle_get_ray(const oracle_camera *cam, double s, double t, double lens_x, double lens_y,
                    double orig[3], double dir[3])
{
    ray r = get_ray(cam, s, t, lens_x, lens_y);
    orig[0] = r.orig.x; orig[1] = r.orig.y; orig[2] = r.orig.z;
    dir[0] = r.dir.x; dir[1] = r.dir.y; dir[2] = r.dir.z;
}

/* The first `count` uniforms of the stream of (pixel, sample), taken as ONE run (unit tests of the word -> uniform rule). */
void oracle_uniforms(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t flags, int32_t count, double *out)
{
    oracle_rng rng;
    rng_init(&rng, seed, pixel, sample);
    rng.u53 = (flags & ORACLE_FLAG_UNIFORM53) != 0;
    if (flags & ORACLE_FLAG_SYMMETRIC_DRAWS) rng_take_sym(&rng, count, out);
    else rng_take(&rng, count, out);
}
