/*
 * oracle_common.h -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * The random stream both oracles (and the HIP kernel, independently
 * implemented in rtiow_amd/csrc) consume.  The reference draws from
 * rand::thread_rng() (main.rs:60,128; vec3.rs:22,27,61; materials.rs:95),
 * which is OS seeded and unreproducible; what is kept is the DISTRIBUTION of
 * each draw and the ORDER of draws.
 *
 * Stream addressing:
 *   key     = (seed lo32, seed hi32)
 *   counter = (pixel index j*W+i, sample index, block index e, 0)
 * The Philox4x32-10 blocks B_0, B_1, ... of one (pixel, sample) are consumed in RUNS of consecutive
 * words (see oracle_rng below): the camera run, one run per Lambertian/Metal scatter, one per
 * Dialectric reflectance draw.
 * A 32-bit word w becomes a draw from [0,1) (gen::<f64>(): main.rs:131-132, materials.rs:96) as
 * u = w * 2^-32, and a draw from the symmetric ranges (gen_range(-1.0..1.0) vec3.rs:63, (-1.0..=1.0)
 * vec3.rs:31-33) as x = (int32_t)w * 2^-31 in [-1,1): the word read as an unsigned resp. a two's-complement
 * integer -- all 32 bits of it either way, exactly representable in f64, so A and B consume identical
 * numbers.  (Rounds 1-4 used u = (w >> 8) * 2^-24 and 2u - 1, a leftover of the abandoned f32 plan.)
 * With ORACLE_FLAG_UNIFORM53 a draw takes TWO consecutive words, u = ((w0 << 32 | w1) >> 11) * 2^-53,
 * and the symmetric ranges are 2u - 1.
 */
#ifndef RTIOW_ORACLE_COMMON_H
#define RTIOW_ORACLE_COMMON_H

#include <stdint.h>
#include "oracle.h"

#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

static inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                 uint32_t k0, uint32_t k1, uint32_t out[4])
{
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0; k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Random words of one (pixel, sample): the Philox blocks B_e = philox(pixel, sample, e, 0), e = 0, 1, ...,
 * consumed in RUNS.  A run takes consecutive words, block after block; when it ends (rng_end_run) the rest of
 * its last block is dropped and the next run starts at a fresh block.  Runs:
 *   camera   B_0 = (u jitter, v jitter, lens x, lens y); every further unit-disk try (vec3.rs:62-63) takes the
 *            next two words -- two tries per block;
 *   a Lambertian/Metal scatter: unit-sphere tries (vec3.rs:31-33) of three consecutive words -- four tries per
 *            three blocks;
 *   a Dialectric's reflectance draw (materials.rs:96): one word, one block, only if can_refract.
 * `explicit_u` (unit tests only) replaces the words by a caller-supplied list of uniforms. */
typedef struct {
    uint32_t k0, k1, pixel, sample, event;
    uint32_t w[4];
    int have;               /* words of the current block not yet taken: w[4 - have .. 3] */
    int u53;                /* ORACLE_FLAG_UNIFORM53: two words per uniform */
    const double *explicit_u;
    int explicit_n, explicit_used;
} oracle_rng;

static inline void rng_init(oracle_rng *r, uint64_t seed, uint32_t pixel, uint32_t sample)
{
    r->k0 = (uint32_t)seed; r->k1 = (uint32_t)(seed >> 32);
    r->pixel = pixel; r->sample = sample; r->event = 0; r->have = 0; r->u53 = 0;
    r->explicit_u = 0; r->explicit_n = 0; r->explicit_used = 0;
}

static inline uint32_t rng_word(oracle_rng *r)
{
    if (r->have == 0) {
        philox4x32_10(r->pixel, r->sample, r->event, 0u, r->k0, r->k1, r->w);
        r->event++;
        r->have = 4;
    }
    return r->w[4 - r->have--];
}

/* Next `count` words of the current run as uniforms w * 2^-32 (exact in f64). */
static inline void rng_take(oracle_rng *r, int count, double *u)
{
    if (r->explicit_u) {
        for (int i = 0; i < count; ++i) {
            u[i] = (r->explicit_used < r->explicit_n) ? r->explicit_u[r->explicit_used] : 0.0;
            r->explicit_used++;
        }
        return;
    }
    for (int i = 0; i < count; ++i) {
        if (r->u53) {       /* 53 bits from two consecutive words (a pair never straddles a block: runs take even counts) */
            const uint64_t hi = rng_word(r), lo = rng_word(r);
            u[i] = (double)(((hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0);
        } else {
            u[i] = (double)rng_word(r) * (1.0 / 4294967296.0);
        }
    }
}
/* Next `count` draws of the current run from the symmetric ranges (-1..1) / (-1..=1): one word each, read as a
 * two's-complement integer, x = (int32_t)w * 2^-31 (exact); 2u - 1 for 53-bit draws and for explicit lists of uniforms. */
static inline void rng_take_sym(oracle_rng *r, int count, double *x)
{
    if (r->explicit_u || r->u53) {
        rng_take(r, count, x);
        for (int i = 0; i < count; ++i) x[i] = 2.0 * x[i] - 1.0;
        return;
    }
    for (int i = 0; i < count; ++i) x[i] = (double)(int32_t)rng_word(r) * (1.0 / 2147483648.0);
}
static inline void rng_end_run(oracle_rng *r) { r->have = 0; }

/* Row list of a call: rows j = row_begin, row_begin+row_step, ... < row_end. */
static inline int params_rows(const oracle_params *p)
{
    int step = p->row_step > 0 ? p->row_step : 1;
    if (p->row_end <= p->row_begin) return 0;
    return (p->row_end - p->row_begin + step - 1) / step;
}

/* Runs fn(arg, row_slot) for row_slot in [0, nrows) on nthreads workers that
 * pull rows from an atomic counter -- the analogue of rayon's into_par_iter
 * over rows (main.rs:122-123).  Defined in oracle_threads.c. */
typedef void (*oracle_row_fn)(void *arg, int row_slot, int worker);
int oracle_parallel_rows(int nrows, int nthreads, oracle_row_fn fn, void *arg);
double oracle_now_seconds(void);

#endif
