// rt_kernels.hpp -- the render megakernel and the resolve kernels (gfx950).
//
// One launch renders a shard of the image: main.rs:122-139 without to_rgba.
//
// Execution shape:
//   * persistent grid: as many 256-thread workgroups as stay resident; every
//     LANE owns one path at a time and never waits for its neighbours -- when
//     its path ends it immediately starts its next pixel-sample ("path
//     regeneration"), so the sphere scan always runs with full waves;
//   * work = items (pixel, chunk of <= `chunk` consecutive samples), handed
//     out by ONE device-wide counter; a wave fetches for all its idle lanes
//     with a single returning atomic (__ballot + popcount + prefix rank);
//   * the sphere scan (HittableList::hit, mod.rs:54-70) walks the list in
//     order with a WAVE-UNIFORM index, so the 16-byte sphere records arrive
//     through the scalar data path (s_load_dwordx4 -> SGPRs) and cost no
//     vector-memory or LDS traffic; 11 VALU ops + 1 compare per test;
//   * tests that reach the square root (sphere.rs:26) are rare per lane but
//     frequent per wave, so they are not evaluated in the scan: the lane
//     appends the sphere index to a short per-lane list in LDS, and the exact
//     root / range logic of sphere.rs:26-34 runs afterwards over that list,
//     in list order (same results);
//   * per-lane radiance sums are exact u64 fixed point (contract C5) and are
//     added to the frame buffer with 64-bit atomics once per item.
#pragma once
#include "rt_device.hpp"

namespace rt {

struct KCamera {
    float origin[3], llc[3], horizontal[3], vertical[3], u[3], v[3];
    float lens_radius;
};

struct KParams {
    KCamera cam;
    int32_t width, height;
    int32_t spp, sample_begin, max_depth;
    float t_min;
    uint32_t k0, k1;
    int32_t tile_rows, shard_index, shard_count;
    int32_t rows;              // compact rows of this shard
    int32_t n_spheres;
    int32_t chunk;             // samples per work item
    uint32_t npix;             // rows * width
    uint32_t total_items;      // npix * ceil(spp / chunk)
    const float4 *geom;        // (cx, cy, cz, r*r)
    const float4 *mat0;        // (1/r, kind bits, param, 0)
    const float4 *mat1;        // (albedo rgb, 0)
    unsigned long long *fix;   // [rows][width][3] exact sums
    unsigned int *queue;       // work counter
    unsigned long long *stats; // [0] rays [1] samples [2] candidates
};

constexpr int RT_KIND_LAMBERTIAN = 0, RT_KIND_METAL = 1, RT_KIND_DIALECTRIC = 2;
constexpr int kBlock = 256;
constexpr int kCandCap = 24;        // per-lane candidate slots
constexpr int kScanUnroll = 8;      // spheres per overflow check

// float4 table read through the constant address space (scalar loads when the
// index is wave-uniform, ordinary vector loads otherwise).
struct ConstF4 {
    const float __attribute__((address_space(4))) *p;
    __device__ __forceinline__ float4 operator[](int i) const
    {
        const float __attribute__((address_space(4))) *q = p + 4 * (size_t)i;
        return make_float4(q[0], q[1], q[2], q[3]);
    }
};

__device__ __forceinline__ V3 ld3(const float *p) { return mk(p[0], p[1], p[2]); }

// sphere.rs:16-24: oc, half_b, c, discriminant for one sphere (contract C2).
__device__ __forceinline__ void sphere_terms(V3 o, V3 d, float a, float4 g, float &half_b, float &disc)
{
    float ocx = o.x - g.x, ocy = o.y - g.y, ocz = o.z - g.z;
    half_b = __builtin_fmaf(ocz, d.z, __builtin_fmaf(ocy, d.y, ocx * d.x));
    float c = __builtin_fmaf(ocz, ocz, __builtin_fmaf(ocy, ocy, __builtin_fmaf(ocx, ocx, -g.w)));
    disc = __builtin_fmaf(half_b, half_b, -(a * c));
}

// sphere.rs:25-34 for one candidate; updates (closest, hit) on acceptance.
__device__ __forceinline__ void sphere_accept(float half_b, float disc, float a, float t_min,
                                              int idx, float &closest, int &hit)
{
    float sqrtd = __builtin_sqrtf(disc);
    float root = (-half_b - sqrtd) / a;
    if (root < t_min || closest < root) {
        root = (-half_b + sqrtd) / a;
        if (root < t_min || closest < root) return;
    }
    closest = root;
    hit = idx;
}

// MODE 0: evaluate sphere.rs:25-34 inline in the scan (simple, divergent).
// MODE 1: deferred candidates through a per-lane LDS list.
template <int MODE>
__global__ __launch_bounds__(kBlock) void render_kernel(const KParams P)
{
    __shared__ uint16_t cand[MODE == 1 ? kCandCap : 1][kBlock];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;

    const V3 cam_origin = ld3(P.cam.origin), cam_llc = ld3(P.cam.llc);
    const V3 cam_hor = ld3(P.cam.horizontal), cam_ver = ld3(P.cam.vertical);
    const V3 cam_u = ld3(P.cam.u), cam_v = ld3(P.cam.v);
    // The sphere list is read-only for the whole launch and indexed by a
    // wave-uniform counter: reading it through the constant address space makes
    // the scan's loads scalar (s_load_dwordx4 into SGPRs).
    const ConstF4 geom{(const float __attribute__((address_space(4))) *)(uintptr_t)P.geom};
    const int n = P.n_spheres;
    const float t_min = P.t_min;
    const float inv_wm1_den = (float)(P.width - 1);
    const float inv_hm1_den = (float)(P.height - 1);

    bool has_item = false, dead = false, alive = false;
    uint32_t pix_local = 0, pix_global = 0;
    int s = 0, s_end = 0;
    float fi = 0.0f, fj = 0.0f;
    unsigned long long acc0 = 0, acc1 = 0, acc2 = 0;
    V3 o = mk(0, 0, 0), d = mk(0, 0, 1), thr = mk(1, 1, 1);
    int depth = 0;
    uint32_t ev = 0;
    uint32_t n_rays = 0, n_samples = 0, n_cand = 0;

    for (;;) {
        // ---- (a) idle lanes fetch a work item: one atomic per wave -----------
        {
            const bool want = !has_item && !dead;
            const unsigned long long m = __ballot(want);
            if (m != 0ull) {
                const uint32_t cnt = (uint32_t)__popcll(m);
                uint32_t base = 0;
                if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(P.queue, cnt);
                base = __shfl(base, (int)__builtin_ctzll(m));
                if (want) {
                    const uint32_t w = base + (uint32_t)__popcll(m & lane_lt);
                    if (w < P.total_items) {
                        const uint32_t c = w / P.npix;
                        pix_local = w - c * P.npix;
                        const uint32_t r = pix_local / (uint32_t)P.width;
                        const uint32_t i = pix_local - r * (uint32_t)P.width;
                        const uint32_t lt = r / (uint32_t)P.tile_rows;             // local tile
                        const uint32_t j = (lt * (uint32_t)P.shard_count + (uint32_t)P.shard_index) * (uint32_t)P.tile_rows
                                           + (r - lt * (uint32_t)P.tile_rows);
                        pix_global = j * (uint32_t)P.width + i;
                        fi = (float)i; fj = (float)j;
                        s = P.sample_begin + (int)c * P.chunk;
                        s_end = min(s + P.chunk, P.sample_begin + P.spp);
                        has_item = true;
                    } else {
                        dead = true;
                    }
                }
            }
        }

        // ---- (b) start the next sample: main.rs:131-134 + camera.rs:47-54 ----
        if (has_item && !alive) {
            U4 w = philox4x32_10(pix_global, (uint32_t)s, 0u, 0u, P.k0, P.k1);
            ev = 1u;
            const float u = (fi + u01(w.x)) / inv_wm1_den;
            const float v = (fj + u01(w.y)) / inv_hm1_den;
            float lx = u11(w.z), ly = u11(w.w);
            while (!(len2(mk(lx, ly, 0.0f)) < 1.0f)) {          // vec3.rs:59-68
                w = philox4x32_10(pix_global, (uint32_t)s, ev, 0u, P.k0, P.k1);
                ev++;
                lx = u11(w.x); ly = u11(w.y);
            }
            const V3 rd = mk(lx, ly, 0.0f) * P.cam.lens_radius;
            const V3 offset = cam_u * rd.x + cam_v * rd.y;
            o = cam_origin + offset;
            d = (((cam_llc + cam_hor * u) + cam_ver * v) - cam_origin) - offset;
            thr = mk(1.0f, 1.0f, 1.0f);
            depth = P.max_depth;
            alive = true;
        }

        // ---- (c) every lane of the wave is out of work: done ------------------
        const unsigned long long alive_mask = __ballot(alive);
        if (alive_mask == 0ull) break;
        n_rays += (uint32_t)__popcll(alive_mask);

        // ---- (d) HittableList::hit, mod.rs:54-70 -------------------------------
        float closest = __builtin_inff();
        int hit = -1;
        if (alive) {
            const float a = len2(d);
            if (MODE == 0) {
                for (int i = 0; i < n; ++i) {
                    const float4 g = geom[i];
                    float half_b, disc;
                    sphere_terms(o, d, a, g, half_b, disc);
                    if (!(disc < 0.0f)) {                       // sphere.rs:25
                        n_cand++;
                        sphere_accept(half_b, disc, a, t_min, i, closest, hit);
                    }
                }
            } else {
                int cnt = 0;
                // exact sphere.rs:25-34 over the lane's pending candidates, in list order
                auto drain = [&]() {
                    // trip count = longest list among the active lanes (exec-masked vote:
                    // no cross-lane data movement inside this divergent region)
                    for (int k = 0; __any(k < cnt); ++k) {
                        if (k < cnt) {
                            const int idx = (int)cand[k][tid];
                            const float4 g = geom[idx];
                            float half_b, disc;
                            sphere_terms(o, d, a, g, half_b, disc);
                            n_cand++;
                            sphere_accept(half_b, disc, a, t_min, idx, closest, hit);
                        }
                    }
                    cnt = 0;
                };
                auto test = [&](int i) {
                    const float4 g = geom[i];
                    float half_b, disc;
                    sphere_terms(o, d, a, g, half_b, disc);
                    if (!(disc < 0.0f)) {                       // sphere.rs:25
                        cand[cnt][tid] = (uint16_t)i;
                        cnt++;
                    }
                };
                const int nfull = n & ~(kScanUnroll - 1);
                for (int i0 = 0; i0 < nfull; i0 += kScanUnroll) {
#pragma unroll
                    for (int k = 0; k < kScanUnroll; ++k) test(i0 + k);
                    // drain before any lane could overflow its list
                    if (__any(cnt > kCandCap - kScanUnroll)) drain();
                }
                for (int i = nfull; i < n; ++i) test(i);
                drain();
            }
        }

        // ---- (e) shade: main.rs:44-56 + materials.rs ----------------------------
        if (alive) {
            bool done = false;
            V3 L = mk(0.0f, 0.0f, 0.0f);
            if (hit < 0) {
                const V3 ud = unit_vector(d);                                   // main.rs:54
                const float t = 0.5f * (ud.y + 1.0f);                           // main.rs:55
                const V3 sky = mk(1.0f, 1.0f, 1.0f) * (1.0f - t) + mk(0.5f, 0.7f, 1.0f) * t;
                L = thr * sky;
                done = true;
            } else {
                const float4 g = geom[hit];
                const float4 m0 = P.mat0[hit];
                const float4 m1 = P.mat1[hit];
                const int kind = __float_as_int(m0.y);
                // sphere.rs:36-37 + mod.rs:20-30 (contract C4)
                const V3 p = mk(__builtin_fmaf(closest, d.x, o.x), __builtin_fmaf(closest, d.y, o.y),
                                __builtin_fmaf(closest, d.z, o.z));
                const V3 outward = (p - mk(g.x, g.y, g.z)) * m0.x;
                const bool front = dot(d, outward) < 0.0f;
                const V3 nrm = front ? outward : (mk(0.0f, 0.0f, 0.0f) - outward);
                V3 ndir = mk(0.0f, 0.0f, 0.0f);
                if (kind != RT_KIND_DIALECTRIC) {
                    V3 sp;
                    do {                                                         // vec3.rs:37-45
                        const U4 w = philox4x32_10(pix_global, (uint32_t)s, ev, 0u, P.k0, P.k1);
                        ev++;
                        sp = mk(u11(w.x), u11(w.y), u11(w.z));
                    } while (!(len2(sp) < 1.0f));
                    if (kind == RT_KIND_LAMBERTIAN) {                            // materials.rs:21-31
                        ndir = nrm + unit_vector(sp);
                        const float eps = 1e-8f;
                        if (__builtin_fabsf(ndir.x) < eps && __builtin_fabsf(ndir.y) < eps &&
                            __builtin_fabsf(ndir.z) < eps)
                            ndir = nrm;
                    } else {                                                     // materials.rs:48-62
                        const V3 reflected = unit_vector(reflect(d, nrm));
                        ndir = reflected + sp * m0.z;
                        if (dot(ndir, nrm) <= 0.0f) done = true;                 // absorbed: L = 0
                    }
                    thr = thr * mk(m1.x, m1.y, m1.z);
                } else {                                                         // materials.rs:76-105
                    const float ratio = front ? 1.0f / m0.z : m0.z;
                    const V3 ud = unit_vector(d);
                    const float cos_theta = min_1(-dot(ud, nrm));
                    const float sin_theta = __builtin_sqrtf(1.0f - cos_theta * cos_theta);
                    bool do_refract = false;
                    if (ratio * sin_theta <= 1.0f) {
                        const float refl = reflectance(cos_theta, ratio);
                        const U4 w = philox4x32_10(pix_global, (uint32_t)s, ev, 0u, P.k0, P.k1);
                        ev++;
                        do_refract = refl <= u01(w.x);
                    }
                    ndir = do_refract ? refract(ud, nrm, ratio) : reflect(ud, nrm);
                    thr = thr * mk(1.0f, 1.0f, 1.0f);
                }
                o = p;
                d = ndir;
                depth -= 1;
                if (depth <= 0) done = true;                                    // main.rs:40-42: L = 0
            }
            if (done) {
                acc0 += quantize(L.x); acc1 += quantize(L.y); acc2 += quantize(L.z);
                alive = false;
                n_samples++;
                s++;
                if (s >= s_end) {
                    unsigned long long *px = P.fix + (size_t)pix_local * 3u;
                    atomicAdd(px + 0, acc0); atomicAdd(px + 1, acc1); atomicAdd(px + 2, acc2);
                    acc0 = acc1 = acc2 = 0ull;
                    has_item = false;
                }
            }
        }
    }

    // wave totals -> device counters
    {
        unsigned long long ns = n_samples, nc = n_cand;
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) { ns += __shfl_xor(ns, sh); nc += __shfl_xor(nc, sh); }
        if (lane == 0) {
            atomicAdd(P.stats + 0, (unsigned long long)n_rays);
            atomicAdd(P.stats + 1, ns);
            atomicAdd(P.stats + 2, nc);
        }
    }
}

// Contract C5: exact sums -> f32 sums.
__global__ void fix_to_f32_kernel(const unsigned long long *__restrict__ fix, float *__restrict__ out, long long count)
{
    long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; k < count; k += stride) {
        const unsigned long long q = fix[k];
        const double dv = (double)(uint32_t)(q >> 32) * 4294967296.0 + (double)(uint32_t)q;
        out[k] = (float)(dv * (1.0 / 4294967296.0));
    }
}

// Color::to_rgba, vec3.rs:403-421, in f32, + the row flip of main.rs:141-145.
__device__ __forceinline__ uint8_t as_u8(float x)
{   // Rust `as u8`: saturating, NaN -> 0
    if (!(x == x)) return 0;
    if (x <= 0.0f) return 0;
    if (x >= 255.0f) return 255;
    return (uint8_t)x;
}
__device__ __forceinline__ float clamp_r(float x, float lo, float hi)
{
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}
__global__ void resolve_rgba8_kernel(const float *__restrict__ sum, uint8_t *__restrict__ out,
                                     int width, int rows, float scale, int flip)
{
    long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long npix = (long long)width * rows;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; k < npix; k += stride) {
        const int r = (int)(k / width);
        const int i = (int)(k - (long long)r * width);
        const int dst = flip ? rows - 1 - r : r;
        const float *c = sum + k * 3;
        uchar4 px;
        px.x = as_u8(256.0f * clamp_r(__builtin_sqrtf(scale * c[0]), 0.0f, 0.999f));
        px.y = as_u8(256.0f * clamp_r(__builtin_sqrtf(scale * c[1]), 0.0f, 0.999f));
        px.z = as_u8(256.0f * clamp_r(__builtin_sqrtf(scale * c[2]), 0.0f, 0.999f));
        px.w = 255;
        reinterpret_cast<uchar4 *>(out)[(long long)dst * width + i] = px;
    }
}

__global__ void philox_kat_kernel(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                  uint32_t k0, uint32_t k1, uint32_t *out)
{
    const U4 r = philox4x32_10(c0, c1, c2, c3, k0, k1);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

} // namespace rt
