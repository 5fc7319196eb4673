/*
 * oracle_threads.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Row-parallel driver (pthread workers + an atomic row counter), Philox KAT
 * export, and Camera::new.
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <time.h>
#include <unistd.h>
#include "oracle_common.h"

void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}

int oracle_hardware_threads(void)
{
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    return n > 0 ? (int)n : 1;
}

double oracle_now_seconds(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

typedef struct {
    atomic_int next;
    int nrows;
    oracle_row_fn fn;
    void *arg;
} pool_t;

typedef struct { pool_t *pool; int worker; } worker_t;

static void *worker_main(void *vp)
{
    worker_t *w = (worker_t *)vp;
    for (;;) {
        int r = atomic_fetch_add(&w->pool->next, 1);
        if (r >= w->pool->nrows) break;
        w->pool->fn(w->pool->arg, r, w->worker);
    }
    return 0;
}

int oracle_parallel_rows(int nrows, int nthreads, oracle_row_fn fn, void *arg)
{
    if (nthreads <= 0) nthreads = oracle_hardware_threads();
    if (nthreads > nrows) nthreads = nrows > 0 ? nrows : 1;
    if (nthreads > 256) nthreads = 256;
    pool_t pool;
    atomic_init(&pool.next, 0);
    pool.nrows = nrows; pool.fn = fn; pool.arg = arg;
    pthread_t tid[256];
    worker_t wk[256];
    for (int i = 0; i < nthreads; ++i) {
        wk[i].pool = &pool; wk[i].worker = i;
        if (i > 0 && pthread_create(&tid[i], 0, worker_main, &wk[i]) != 0) return -1;
    }
    worker_main(&wk[0]);
    for (int i = 1; i < nthreads; ++i) pthread_join(tid[i], 0);
    return nthreads;
}

/* ---- Camera::new, camera.rs:17-45 (host-only setup, f64) ---------------- */

static void v_sub(const double a[3], const double b[3], double o[3]) { for (int i = 0; i < 3; ++i) o[i] = a[i] - b[i]; }
static void v_scale(const double a[3], double s, double o[3]) { for (int i = 0; i < 3; ++i) o[i] = a[i] * s; }
static void v_cross(const double a[3], const double b[3], double o[3])
{   /* vec3.rs:99-105 */
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
static void v_unit(const double a[3], double o[3])
{   /* vec3.rs:107-109 with Div<f64> = mul by reciprocal (vec3.rs:371-375) */
    double len = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    v_scale(a, 1.0 / len, o);
}

void oracle_camera_new(const double look_from[3], const double look_at[3], const double v_up[3],
                       double v_fov_deg, double aspect_ratio, double aperture, double focus_dist,
                       oracle_camera *out)
{
    /* f64::to_radians multiplies by pi/180 (camera.rs:25). */
    double theta = v_fov_deg * (3.14159265358979323846 / 180.0);
    double viewport_height = 2.0 * tan(theta / 2.0);
    double viewport_width = aspect_ratio * viewport_height;
    double w[3], u[3], v[3], t[3];
    v_sub(look_from, look_at, t); v_unit(t, w);
    v_cross(v_up, w, t); v_unit(t, u);
    v_cross(w, u, v);
    /* focus_dist * viewport_width * u : (f64*f64)*Vec3, camera.rs:33-34 */
    v_scale(u, focus_dist * viewport_width, out->horizontal);
    v_scale(v, focus_dist * viewport_height, out->vertical);
    double h2[3], v2[3], fw[3];
    v_scale(out->horizontal, 1.0 / 2.0, h2);     /* Vec3 / 2.0 = * (1/2) */
    v_scale(out->vertical, 1.0 / 2.0, v2);
    v_scale(w, focus_dist, fw);
    for (int i = 0; i < 3; ++i) {
        out->origin[i] = look_from[i];
        out->lower_left_corner[i] = ((look_from[i] - h2[i]) - v2[i]) - fw[i];
        out->u[i] = u[i]; out->v[i] = v[i];
    }
    out->lens_radius = aperture / 2.0;
}

