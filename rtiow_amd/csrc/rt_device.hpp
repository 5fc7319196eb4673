// rt_device.hpp -- device-side arithmetic of the render megakernel (gfx950).
//
// Two kinds of arithmetic live here (DESIGN.md sections 4 and 5.2):
//
//  * the EXACT path, in IEEE binary64, in the reference's operation order and
//    with no fused multiply-add (this translation unit is compiled with
//    -ffp-contract=off and no f64 fma is ever written out).  It decides every
//    hit and computes every shading value; it is written to agree with
//    oracle/oracle_f64.c (Oracle B) bit for bit.  / and sqrt are the correctly
//    rounded f64 expansions of the compiler.
//  * the f32 FILTER of the sphere scan, which only ever answers "this sphere
//    cannot be hit" or "ask the exact path".
//
// Reference lines cited are paths under the upstream repo (src/...).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

struct D3 { double x, y, z; };

__device__ __forceinline__ D3 mk(double x, double y, double z) { D3 r; r.x = x; r.y = y; r.z = z; return r; }
// vec3.rs:137-147, :243-253, :330-356, :358-367
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ D3 operator*(D3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ D3 operator*(D3 a, D3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }

// vec3.rs:95-97 and :87-89 (powi(2) = x*x): left-to-right sums of rounded products.
__device__ __forceinline__ double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ double length_squared(D3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
// vec3.rs:107-109 with Div<f64> = multiply by the reciprocal (:371-375).
__device__ __forceinline__ D3 unit_vector(D3 a) { return a * (1.0 / __builtin_sqrt(length_squared(a))); }
// vec3.rs:116-118:  v - (2*dot(v,n)) * n
__device__ __forceinline__ D3 reflect(D3 v, D3 n) { return v - n * (2.0 * dot(v, n)); }
__device__ __forceinline__ double min_1(double x) { return (x < 1.0) ? x : 1.0; }   // 1.0.min(x): NaN -> 1

// vec3.rs:120-125
__device__ __forceinline__ D3 refract(D3 uv, D3 n, double etai_over_etat)
{
    const double cos_theta = min_1(-dot(uv, n));
    const D3 perp = (uv + n * cos_theta) * etai_over_etat;
    const D3 par = n * (-__builtin_sqrt(__builtin_fabs(1.0 - length_squared(perp))));
    return perp + par;
}

// materials.rs:78-82; powi(2) = x*x, powi(5) = ((x*x)*(x*x))*x
__device__ __forceinline__ double reflectance(double cosine, double ref_idx)
{
    double r0 = (1.0 - ref_idx) / (1.0 + ref_idx);
    r0 = r0 * r0;
    const double x = 1.0 - cosine;
    const double x2 = x * x;
    const double x5 = (x2 * x2) * x;
    return r0 + (1.0 - r0) * x5;
}

// ---- Philox4x32-10 (Random123), counter = (pixel, sample, event, 0) ---------
struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1)
{
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    constexpr uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    // The key is wave-uniform, and so are the ten round keys: the compiler would compute them once
    // per kernel and keep 20 SGPRs alive across the whole bounce loop (and spill others to make room).
    // Making the key opaque here has every call rebuild them with 18 scalar adds instead.
    asm volatile("" : "+s"(k0), "+s"(k1));
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c0;      // hi and lo of one 32x32 product
        const uint64_t p1 = (uint64_t)M1 * c2;
        // hi ^ counter ^ key as ONE v_bitop3_b32 (truth table 0x96 = three-input XOR; gfx950 has no v_xor3 and the
        // compiler emits two v_xor_b32 for the plain expression: 40 of a block's ~57 vector instructions were XORs)
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += W0; k1 += W1;
    }
    U4 o; o.x = c0; o.y = c1; o.z = c2; o.w = c3;
    return o;
}

// word -> draw from [0,1) (gen::<f64>()): u = w * 2^-32, all 32 bits of the word (exact in f64).  (Rounds 1-4: (w >> 8) * 2^-24,
// a leftover of the abandoned f32 plan; the shift is gone and the lattice is 256 times finer for the same Philox work.)
__device__ __forceinline__ double u01(uint32_t w) { return (double)w * (1.0 / 4294967296.0); }
// word -> draw from the symmetric ranges gen_range(-1.0..1.0) / (-1.0..=1.0): the word read as a TWO'S-COMPLEMENT integer,
// x = m * 2^-31 with m = (int32_t)w in [-2^31, 2^31): exact, two instructions (v_cvt_f64_i32, v_mul_f64), and the integer m is what
// the rejection tests below square -- no bias to subtract first.  (As uniform on its lattice as 2 u01(w) - 1, which is this value
// for the word with its top bit flipped.)
__device__ __forceinline__ int32_t u11_int(uint32_t w) { return (int32_t)w; }
__device__ __forceinline__ double u11(uint32_t w) { return (double)u11_int(w) * (1.0 / 2147483648.0); }

// RT_FLAG_UNIFORM53: a uniform from TWO consecutive words, u = ((w0 << 32 | w1) >> 11) * 2^-53 -- the 53 random bits of
// rand 0.8.5's gen::<f64>() (main.rs:131-132, materials.rs:96).  k = w0 * 2^21 + (w1 >> 11) < 2^53: both conversions, the
// sum and the scaling are exact.  Ranges (-1..1) and (-1..=1) map to 2u - 1 (exact: an even integer of at most 54 bits / 2^53).
__device__ __forceinline__ double u01_53(uint32_t w0, uint32_t w1)
{
    return ((double)w0 * 2097152.0 + (double)(w1 >> 11)) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ double u11_53(uint32_t w0, uint32_t w1) { return 2.0 * u01_53(w0, w1) - 1.0; }

// The reference's rejection tests on such uniforms (vec3.rs:37-45 `length_squared() < 1.0`, vec3.rs:59-68), bit for bit, on the
// integers.  x_i = u11(w_i) = m_i * 2^-31 with |m_i| <= 2^31, so the exact sum of squares is S * 2^-62 with the integer
// S = sum m_i^2 <= 3 * 2^62 < 2^64 (three v_mad_i64_i32).  The reference evaluates x*x + y*y + z*z in f64, rounding each product
// and each sum (m_i^2 has up to 62 bits): its value s differs from the exact one by at most 3 * 2^-53 * S * 2^-62 * (1 + 2^-52) < 2^-49
// (S * 2^-62 <= 3).  Hence
//     S <  2^62 - 2^32   =>  exact <= 1 - 2^-30  =>  s < 1.0        (accepted, as the reference accepts)
//     S >= 2^62 + 2^32   =>  exact >= 1 + 2^-30  =>  s > 1.0        (rejected, as the reference rejects)
// and only in between -- the high dword of S is 2^30 - 1 or 2^30: 1.5e-9 of the tries -- can the roundings decide; there the
// expression is evaluated in f64 exactly as the reference writes it (vec3.rs:87-89: (x*x + y*y) + z*z, no fused multiply-add).
// The retry loops run on the integers and convert the accepted draw once.  (Rounds 1-4: a 2^-23 lattice, where every product and
// sum was exact in f64 and no fallback was needed.)
__device__ __forceinline__ bool unit_norm_accepts_slow(uint32_t wx, uint32_t wy, uint32_t wz)
{
    const double x = u11(wx), y = u11(wy), z = u11(wz);
    return x * x + y * y + z * z < 1.0;
}
__device__ __forceinline__ bool unit_disk_accepts(uint32_t wx, uint32_t wy)
{
    const int32_t x = u11_int(wx), y = u11_int(wy);
    const unsigned long long S = (unsigned long long)((long long)x * x) + (unsigned long long)((long long)y * y);
    const uint32_t hi = (uint32_t)(S >> 32);
    bool acc = hi < 0x3FFFFFFFu;
    // (vec3.rs:65 tests Vec3::new(x, y, 0.0).length_squared(): (x*x + y*y) + 0.0*0.0 -- the word 0 gives z = 0.0)
    if (__builtin_expect(hi - 0x3FFFFFFFu < 2u, 0)) acc = unit_norm_accepts_slow(wx, wy, 0u);
    return acc;
}
__device__ __forceinline__ bool unit_sphere_accepts(uint32_t wx, uint32_t wy, uint32_t wz)
{
    const int32_t x = u11_int(wx), y = u11_int(wy), z = u11_int(wz);
    const unsigned long long S = (unsigned long long)((long long)x * x) + (unsigned long long)((long long)y * y)
                                 + (unsigned long long)((long long)z * z);
    const uint32_t hi = (uint32_t)(S >> 32);
    bool acc = hi < 0x3FFFFFFFu;
    if (__builtin_expect(hi - 0x3FFFFFFFu < 2u, 0)) acc = unit_norm_accepts_slow(wx, wy, wz);
    return acc;
}

// Contract C5: truncate one radiance channel to the 2^-32 grid, clamped at 2^16 = kSampleClamp (include/rtiow_hip.h,
// RT_SAMPLE_CLAMP: with q <= 2^48 a pixel's u64 sum cannot wrap below 65 536 samples, and below 65 536 samples a clamped
// sample alone puts the pixel's mean at >= 1, i.e. at byte 255, where the reference's unbounded f64 sum puts it too).
// (= (unsigned long long)(x * 2^32) for 0 <= x <= 2^16, 0 for NaN and negatives, 2^48 above -- written so that it
//  compiles to 7 instructions instead of the 13 of a generic f64 -> u64 conversion: the integer part converts
//  exactly (v_cvt_u32_f64 truncates), the rest x - hi is exact, and so is its product with 2^32)
__device__ __forceinline__ unsigned long long quantize(double x)
{
    x = __builtin_fmin(__builtin_fmax(x, 0.0), 65536.0);            // NaN -> 0: maxNum(NaN, 0) is 0 (the max FIRST)
    const uint32_t hi = (uint32_t)x;
    const uint32_t lo = (uint32_t)((x - (double)hi) * 4294967296.0);
    return ((unsigned long long)hi << 32) | (unsigned long long)lo;
}

// ---- the f32 filter of the sphere scan ----------------
//
// For a ray (o, d) and a sphere (c, r) the reference rejects when
//     disc = half_b^2 - a*cc < 0,  half_b = oc.d, a = d.d, cc = oc.oc - r^2  (sphere.rs:18-25),
// whose sign does not depend on the length of d.  With g = d / sqrt(a (1 - KU)), so that
// |g|^2 = 1/(1-KU), kappa = KU/(1-KU), and the per-sphere constant
//     K' <= |c|^2 (1 - kappa) - r^2 (1 + 2 kappa)
// the filter evaluates, in f32 with fused multiply-adds,
//     hb  = o.g - c.g                                   (3 fma, o.g once per ray)
//     D'' = hb^2 - ( |o|^2 (1 - kappa) - 2 o.c )        (3 fma + 1 fma)
// and keeps the sphere iff D'' >= K'.  In exact arithmetic
//     D'' - K' = disc/a + kappa ( (oc.d)^2/a + 2 r^2 + |c|^2 + |o|^2 )  >=  disc/a + kappa S,
// S = |o|^2 + |c|^2 + r^2.  The f32 evaluation is off by at most 61 u S, u = 2^-24 (error
// analysis in DESIGN.md section 5.2); kappa > KU = 128 u.  Hence disc >= 0 implies the
// sphere is kept: a sphere the filter drops is one the reference rejects.
// KU per evaluation scheme: 128 u for the f32 fma chains (VALU, f32 MFMA); 1024 u for the
// bf16x3 matrix form, whose 32-term accumulation is bounded by 472 u S.
constexpr float kUnitRoundoff = 5.9604644775390625e-08f;                // 2^-24
constexpr float kFilterKU = 128.0f * kUnitRoundoff;                     // 2^-17
constexpr float kFilterKU_bf16x3 = 1024.0f * kUnitRoundoff;             // 2^-14

struct RayFilter {
    float gx, gy, gz;       // d / sqrt(a (1-KU))
    float h0;               // o.g   (+inf: ray outside the analysed range -> keep everything)
    float px, py, pz;       // -2 o
    float o2;               // |o|^2 (1 - kappa)
    bool sane;              // false: outside the analysed range, every sphere must be kept
};

template <bool BF16X3 = false>
__device__ __forceinline__ RayFilter make_filter(D3 o, D3 d)
{
    constexpr float KU = BF16X3 ? kFilterKU_bf16x3 : kFilterKU;
    RayFilter f;
    const float ox = (float)o.x, oy = (float)o.y, oz = (float)o.z;
    const float dx = (float)d.x, dy = (float)d.y, dz = (float)d.z;
    const float a = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    const float oo = __builtin_fmaf(oz, oz, __builtin_fmaf(oy, oy, ox * ox));
    const float s = __builtin_amdgcn_rsqf(a * (1.0f - KU));                 // v_rsq_f32, 1 ulp
    f.gx = dx * s; f.gy = dy * s; f.gz = dz * s;
    f.h0 = __builtin_fmaf(oz, f.gz, __builtin_fmaf(oy, f.gy, ox * f.gx));
    f.px = -2.0f * ox; f.py = -2.0f * oy; f.pz = -2.0f * oz;
    f.o2 = oo * (1.0f - KU / (1.0f - KU));
    // outside the range where the relative-error analysis holds: let everything through
    f.sane = (a > 1e-20f) && (a < 1e20f) && (oo < 1e30f);
    if (!f.sane) {
        f.gx = f.gy = f.gz = 0.0f; f.px = f.py = f.pz = 0.0f; f.o2 = 0.0f;
        f.h0 = BF16X3 ? 0.0f : __builtin_inff();                            // f32 forms: D'' = +inf
    }
    return f;
}

// true: the sphere is kept for the exact test.  (cx, cy, cz, kp) is its filter record.
__device__ __forceinline__ bool filter_keeps(const RayFilter &f, float cx, float cy, float cz, float kp)
{
    const float hb = __builtin_fmaf(-cx, f.gx, __builtin_fmaf(-cy, f.gy, __builtin_fmaf(-cz, f.gz, f.h0)));
    const float q = __builtin_fmaf(cx, f.px, __builtin_fmaf(cy, f.py, __builtin_fmaf(cz, f.pz, f.o2)));
    const float D = __builtin_fmaf(hb, hb, -q);
    return D >= kp;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// (row stride, in uint4, of the operand staging of scan mode 4 -- xcheck/rt_xcheck_device.hpp; named here because a static_assert of
//  render_kernel sizes the shared bitmap area against it in every instantiation)
constexpr int kStageStride = 9;

// ---- MODE 5: the tube filter ------------------------------------------------------------------
// A ray can only hit sphere (c, r) if its LINE passes within r of c, i.e. |(c-o)_perp| <= r.  For any
// two orthonormal directions u1, u2 perpendicular to the ray that implies |u_k.(c-o)| <= r, k = 1, 2:
// the sphere lies in a square tube around the line.  Each h_k = u_k.c - u_k.o is LINEAR in c, so its
// rounding error is relative to |c| and |o|, not to their squares: two bf16 pieces per factor are
// enough (error ~ 3e-5 |c|, 0.3 % of the book scene's radii; the quadratic forms of modes 1-4 need
// three pieces and still carry a slack of 6e-5 (|o|^2+|c|^2)).  Per (ray, direction) row and
// per-sphere column the 16 K-slots of one v_mfma_f32_32x32x16_bf16 are
//
//   slots 4i .. 4i+3 (i = x,y,z):  A (x1,x1,x2,x2)  B (y1,y2,y1,y2)   = (x1+x2)(y1+y2),  y = sigma c_i
//   slots 12,13,14:                A (t1,t2,t3)     B (sigma x 3)     t = -(u_k.o), split exactly; sigma a bf16
//   slot 15:                       A 1              B 0 (4 in a column that is never kept)
//
// and one instruction evaluates 16 rays x 2 directions against 32 spheres.  Soundness:
// with the basis errors (|u_k.d^| <= 64u, ||u_k|-1| <= 64u, measured <= 8u), the operand truncation
// (2^-16 per factor) and the accumulation (charged 2u per add on sum|terms|),
//     hit  =>  |h_k| <= R + e,   R = r (1+64u) + 640u |c|   (per sphere),   e = 128u |o|_1   (per ray; the 1-norm
//     |ox| + |oy| + |oz| >= |o| is two additions where the 2-norm is a correctly rounded f32 square root: 17 instructions).
// The per-ray part is folded into the rows: they are scaled by lambda = rho / (rho + e), which makes
// lambda |h_k| <= max(R, rho) for every sphere (rho: a per-scene radius floor chosen on the host).
// The host folds each sphere's bound into its column: the column holds sigma c and sigma instead of c and 1, with
// sigma = 2 (1 - 2^-6) / max(R, rho) rounded DOWN to a bf16, so that
//     hit  =>  |H_k| = sigma lambda |h_k| <= 2 (1 - 2^-6) < 2 = kTubeKeepBelow,
// and "kept" is a test of ONE BIT of the f32 pattern (biased exponent < 128).  Every error term of the budget above is
// relative to |c| or |o| and scales with sigma like h itself; the t slots multiply the exact pieces of t by the exact
// bf16 sigma; 2^-6 covers the rounding of the f32 result next to 2.  K-slot 15 carries 1 (A side) x 0 or 4 (B side):
// a column that must never be kept (padding, spheres on the always-exact list) is all zero but for that 4.
constexpr float kTubeKeepBelow = 2.0f;
constexpr float kTubeBasisErr = 64.0f * kUnitRoundoff;
constexpr float kTubeCenterErr = 640.0f * kUnitRoundoff;
constexpr float kTubeOriginErr = 128.0f * kUnitRoundoff;

// MODE 5 tiles its spheres by position: a square grid of G x G cells over x in [g0, g3], z in [g1, g4] (cell size
// 1 / g2), and every sphere that lives in a cell has its centre in that cell, radius <= g7 and its whole extent within
// y in [g5, g6] (rt_api.hip).  Which cells can hold a sphere the ray o + t d, t > 0, hits?  The hit point lies on the
// ray, inside the slab g5 <= y <= g6 and within g7 of the sphere's centre in x and z: so clip the ray to the box
// [g0 - g7, g3 + g7] x [g5, g6] x [g1 - g7, g4 + g7], take the xz bounding rectangle of the clipped piece, grow it by
// g7 and return the cells it overlaps: columns ix0 .. ix0 + nx - 1, rows iz0 .. iz0 + nz - 1; the return value is
// nx * nz, 0 when the ray misses the box, -1 when the question cannot be answered (the wave then scans every tile).
// f32 arithmetic, biased to include:
//   * o and d are rounded to f32 (6e-8 relative each): the line they define stays within 6e-8 (|o| + L) of the true
//     one at distance L, and a sphere can only matter for L <= |o| + the scene's size (`scale`: the sum over the axes
//     of the largest |coordinate| the grid's box reaches), so the box is grown by e = 1e-6 (|o|_1 + scale) on every side;
//   * v_rcp_f32 and the products put <= 3e-7 relative error on each slab parameter: the interval test calls the
//     clipped piece empty only if the exit parameter lies below 0.9999 of the entry parameter; the two end points
//     carry that error on a length <= |o| + scale, and their own rounding: the rectangle is grown by g7 + 4 e;
//   * the cell coordinates are rounded 1e-3 cells outwards;
//   * a direction component of magnitude < 1e-30, |o| or |d| beyond 1e15, an end point beyond 1e30 or a NaN: "cannot
//     tell" (-1).
// The same footprint ROW BY ROW (large grids): the bounding rectangle of a long diagonal piece holds far more cells
// than the line crosses (a ray that leaves a sphere nearly horizontally stays inside the slab up to the box's edge: 3 %
// of the rays of the 10k-sphere scene, and their rectangles made 12 cells per wave-pass out of 5).  GridSeg keeps the
// clipped piece in CELL coordinates; grid_row_run() gives the columns of one grid row: the x range of the part of the
// piece whose z lies in that row's band.  Margins: the f32 end points lie within 4 e of the true ones, so every point of
// the true piece is within 6 e of the computed segment (as point sets -- no amplification by the slope); the band and the
// run are grown by pad + 10 e (in cells) + 1e-3 cells, the run by 1e-2 cells more for the slope's own rounding; a piece
// whose z extent is below 1e-2 cells (the slope would be ill-conditioned) or whose slope is not finite takes the
// rectangle's whole run; a run never leaves the rectangle's columns (it is clamped to them as cells).  The band's ends
// are clipped to the piece BEFORE x is evaluated (the differences from end a stay below the piece's own extent: a steep
// slope multiplies no cancellation error), and their rounding (<= 8e-6 cells) is inside the band's 1e-3.
struct GridSeg {
    // The piece in the coordinate zeta = sg * z (sg = the sign of the slope dx/dz, so that x INCREASES with zeta), relative to
    // end a: x = Xa + (zeta - zeta_a) * |SL|.  Everything a row needs is folded into eight per-ray constants, so one row is
    // two fma for the band's ends, one max and one min to clip them to the piece, two fma for x (15 vector instructions with
    // the conversions and clamps; the form with z clipped first, both end points evaluated and sorted, and the rectangle
    // clipped afterwards took 32):
    float sg;                   // +1 or -1
    float alo, ahi;             // band of row iz in zeta, relative to end a: [sg * iz + alo, sg * iz + ahi]  (margin m inside)
    float dmin, dmax;           // the piece's zeta range relative to end a (one of them is 0)
    float sl;                   // |SL|; 0 for a piece that takes the rectangle's whole run
    float xl, xh;               // Xa -+ (1e-2 + m); a whole-run piece: Xlo - m, Xhi + m
};
// (c0 .. c1: the rectangle's columns, grid_cells' ix0 .. ix0 + nx - 1; fz: the row as a float)
__device__ __forceinline__ void grid_row_run(const GridSeg &s, int c0, int c1, float fz, int &ix0, int &nx)
{
    const float dl = __builtin_fmaxf(__builtin_fmaf(s.sg, fz, s.alo), s.dmin), dh = __builtin_fminf(__builtin_fmaf(s.sg, fz, s.ahi), s.dmax);
    const float lo = __builtin_fmaf(dl, s.sl, s.xl), hi = __builtin_fmaf(dh, s.sl, s.xh);
    ix0 = min(max((int)__builtin_floorf(lo), c0), c1);
    // (nx >= 1 by construction: the band always overlaps the piece, dl <= dh, by margins of 6e cells against ~8e-6 cells of rounding --
    //  but a run of 0 or fewer columns would make the caller's mask shift undefined and the filter unsound, so it is also structural)
    nx = max(min(max((int)__builtin_floorf(hi), c0), c1) - ix0, 0) + 1;
}

// (of, df: the ray's origin and direction rounded to f32, o1 = |of|_1 -- shared with make_tube)
__device__ __forceinline__ int grid_cells(const float (&of)[3], const float (&df)[3], float o1, const float (&g)[8], int G, float scale,
                                          int &ix0, int &nx, int &iz0, int &nz, GridSeg *seg = nullptr)
{
    const float e = 1e-6f * (o1 + scale);
    const float dmin = __builtin_fminf(__builtin_fminf(__builtin_fabsf(df[0]), __builtin_fabsf(df[1])), __builtin_fabsf(df[2]));
    const float dmax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(df[0]), __builtin_fabsf(df[1])), __builtin_fabsf(df[2]));
    // ONE exit: everything below is computed whatever the answer (garbage in, clamped garbage out) and the verdict is
    // selected at the end -- four early returns cost five register moves each
    const bool cannot0 = !(dmin > 1e-30f && dmax < 1e15f && o1 < 1e15f);
    const float m = g[7] + e;
    const float lo[3] = {g[0] - m, g[5] - e, g[1] - m}, hi[3] = {g[3] + m, g[6] + e, g[4] + m};
    float t_in = 0.0f, t_out = __builtin_inff();
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float inv = __builtin_amdgcn_rcpf(df[a]);
        const float t0 = (lo[a] - of[a]) * inv, t1 = (hi[a] - of[a]) * inv;
        t_in = __builtin_fmaxf(t_in, __builtin_fminf(t0, t1));
        t_out = __builtin_fminf(t_out, __builtin_fmaxf(t0, t1));
    }
    const bool empty = t_out < t_in * 0.9999f;
    bool cannot1 = !(t_out < 1e30f);
    const float m2 = g[7] + 4.0f * e;
    const float xa = __builtin_fmaf(t_in, df[0], of[0]), xb = __builtin_fmaf(t_out, df[0], of[0]);
    const float za = __builtin_fmaf(t_in, df[2], of[2]), zb = __builtin_fmaf(t_out, df[2], of[2]);
    const float fx0 = ((__builtin_fminf(xa, xb) - m2) - g[0]) * g[2] - 1e-3f, fx1 = ((__builtin_fmaxf(xa, xb) + m2) - g[0]) * g[2] + 1e-3f;
    const float fz0 = ((__builtin_fminf(za, zb) - m2) - g[1]) * g[2] - 1e-3f, fz1 = ((__builtin_fmaxf(za, zb) + m2) - g[1]) * g[2] + 1e-3f;
    cannot1 = cannot1 || !(fx0 <= fx1 && fz0 <= fz1);                                   // (a NaN)
    if (seg) {
        const float Xa = (xa - g[0]) * g[2], Xb = (xb - g[0]) * g[2], Za = (za - g[1]) * g[2], Zb = (zb - g[1]) * g[2];
        const float m = (g[7] + 10.0f * e) * g[2] + 1e-3f;
        const float dz = Zb - Za, SL = (Xb - Xa) * __builtin_amdgcn_rcpf(dz);
        const bool whole = !(__builtin_fabsf(dz) >= 1e-2f && __builtin_fabsf(SL) < 1e6f);     // (too flat for a slope, or not finite)
        const bool neg = !whole && SL < 0.0f;
        const float za_ = neg ? -Za : Za, zb_ = neg ? -Zb : Zb;                                 // zeta of the two ends
        seg->sg = neg ? -1.0f : 1.0f;
        seg->alo = (neg ? -(1.0f + m) : -m) - za_;
        seg->ahi = (neg ? m : 1.0f + m) - za_;
        seg->dmin = __builtin_fminf(zb_ - za_, 0.0f);
        seg->dmax = __builtin_fmaxf(zb_ - za_, 0.0f);
        seg->sl = whole ? 0.0f : __builtin_fabsf(SL);
        seg->xl = (whole ? __builtin_fminf(Xa, Xb) : Xa - 1e-2f) - m;
        seg->xh = (whole ? __builtin_fmaxf(Xa, Xb) : Xa + 1e-2f) + m;
    }
    // (float -> int conversions saturate, a NaN converts to 0; the cells are clamped to the grid like the host clamps the centres)
    ix0 = min(max((int)__builtin_floorf(fx0), 0), G - 1);
    iz0 = min(max((int)__builtin_floorf(fz0), 0), G - 1);
    nx = min(max((int)__builtin_floorf(fx1), 0), G - 1) - ix0 + 1;
    nz = min(max((int)__builtin_floorf(fz1), 0), G - 1) - iz0 + 1;
    return cannot0 ? -1 : (empty ? 0 : (cannot1 ? -1 : nx * nz));          // (the order of the four early returns this replaces)
}

// the ray as the f32 filter and the footprint see it: six conversions and the 1-norm of the origin, once per bounce
__device__ __forceinline__ void ray_f32(D3 o, D3 d, float (&of)[3], float (&df)[3], float &o1)
{
    of[0] = (float)o.x; of[1] = (float)o.y; of[2] = (float)o.z;
    df[0] = (float)d.x; df[1] = (float)d.y; df[2] = (float)d.z;
    o1 = (__builtin_fabsf(of[0]) + __builtin_fabsf(of[1])) + __builtin_fabsf(of[2]);
}
struct TubeRay {
    float u[2][3];      // lambda * u_k
    float t[2];         // -(lambda u_k) . o
    bool sane;          // false: outside the analysed range, every sphere must be tested exactly
};

// (of, df: the ray's origin and direction rounded to f32; o1 = |of|_1 >= |of|: ray_f32() below)
// SANITIZE = false: the caller replaces the rows of a ray outside the analysed range itself (tube_rows_keep_nothing) -- the render kernel
// does it behind ONE wave-level branch together with the lanes that have no ray: eight selects per pass that nearly never select anything
template <bool SANITIZE = true>
__device__ __forceinline__ TubeRay make_tube(const float (&of)[3], const float (&df)[3], float o1, float rho)
{
    TubeRay T;
    const float ox = of[0], oy = of[1], oz = of[2];
    const float dx = df[0], dy = df[1], dz = df[2];
    const float a = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    const float s = __builtin_amdgcn_rsqf(a);
    const float gx = dx * s, gy = dy * s, gz = dz * s;
    // orthonormal basis without a branch or a singular direction (Duff et al., JCGT 2017)
    const float sg = __builtin_copysignf(1.0f, gz);
    const float aa = -__builtin_amdgcn_rcpf(sg + gz);
    const float b = gx * gy * aa;
    const float lam = rho * __builtin_amdgcn_rcpf(__builtin_fmaf(kTubeOriginErr, o1, rho));
    T.u[0][0] = lam * __builtin_fmaf(sg * gx, gx * aa, 1.0f);
    T.u[0][1] = lam * (sg * b);
    T.u[0][2] = lam * (-sg * gx);
    T.u[1][0] = lam * b;
    T.u[1][1] = lam * __builtin_fmaf(gy, gy * aa, sg);
    T.u[1][2] = lam * (-gy);
#pragma unroll
    for (int k = 0; k < 2; ++k)
        T.t[k] = -__builtin_fmaf(T.u[k][2], oz, __builtin_fmaf(T.u[k][1], oy, T.u[k][0] * ox));
    T.sane = (a > 1e-20f) && (a < 1e20f) && (o1 < 1e15f);           // (|o| <= |o|_1 < 1e15: |o|^2 < 1e30, the analysed range)
    if constexpr (SANITIZE) {
        if (!T.sane) {
#pragma unroll
            for (int k = 0; k < 2; ++k) { T.u[k][0] = T.u[k][1] = T.u[k][2] = 0.0f; T.t[k] = 3.0e38f; }
        }
    }
    return T;
}
// the rows of a ray that keeps NO column: |h| is huge everywhere (a lane without a ray; a ray outside the analysed range, which is tested
// exactly against the whole list instead)
__device__ __forceinline__ void tube_rows_keep_nothing(TubeRay &T)
{
#pragma unroll
    for (int k = 0; k < 2; ++k) { T.u[k][0] = T.u[k][1] = T.u[k][2] = 0.0f; T.t[k] = 3.0e38f; }
}
__device__ __forceinline__ TubeRay make_tube(D3 o, D3 d, float rho)
{
    float of[3], df[3], o1;
    ray_f32(o, d, of, df, o1);
    return make_tube(of, df, o1, rho);
}
// a lane without a ray: |h| is huge for every column
__device__ __forceinline__ TubeRay no_tube_ray()
{
    TubeRay T;
#pragma unroll
    for (int k = 0; k < 2; ++k) { T.u[k][0] = T.u[k][1] = T.u[k][2] = 0.0f; T.t[k] = 3.0e38f; }
    T.sane = true;
    return T;
}

// the two rows of a ray as 2 x 8 dwords (two bf16 each, low half = even K-slot).  The pieces are the hardware's
// round-to-nearest-even conversions, two per instruction (v_cvt_pk_bf16_f32): converting (x, x) gives the dword
// p1 | p1 << 16 at once, and its high half IS the f32 value of the piece.
__device__ __forceinline__ void tube_a_words(const TubeRay &T, uint32_t (&w)[2][8])
{
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
    auto pk = [](float lo, float hi) -> uint32_t {
        const f32x2v v = {lo, hi};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2v));
    };
#pragma unroll
    for (int k = 0; k < 2; ++k) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float x = T.u[k][i];
            const uint32_t d1 = pk(x, x);                                       // p1 | p1 << 16
            const float r1 = x - __uint_as_float(d1 & 0xFFFF0000u);
            w[k][2 * i + 0] = d1;
            w[k][2 * i + 1] = pk(r1, r1);                                       // p2 | p2 << 16
        }
        // t in three pieces (the third is exact: at most 8 significant bits are left), then the 1 of K-slot 15
        // (pk(0, t) IS the f32 value of t's first piece, p1 << 16, and pk(t, r1) the dword p1 | p2 << 16 in one conversion -- the low half converts
        //  t again: six instructions where converting (t, t) and (r1, r1) and masking and merging the halves took eight)
        const float t = T.t[k];
        const float r1 = t - __uint_as_float(pk(0.0f, t));
        const uint32_t c12 = pk(t, r1);                                         // p1 | p2 << 16
        const float r2 = r1 - __uint_as_float(c12 & 0xFFFF0000u);
        w[k][6] = c12;
        w[k][7] = pk(r2, 1.0f);                     // p3 | 1 << 16; slot 15: 1 (against 0, or 4 in a never-kept column)
    }
}

// Per-ray dwords -> MFMA A operands through LDS (4 KB for the wave's 64 rays, one round).  For
// v_mfma_f32_32x32x16_bf16 lane l holds row l&31 and K-slots 8(l>>5) .. +7; rows are laid out so
// that a result lane owns BOTH directions of its rays: row 8b + x (x = 0..7) = ray 8(b>>1) + x,
// direction b&1.  The 16-byte quarter q = 2 dir + khalf of ray r sits at r*4 + (q ^ ((r>>1)&3)):
// conflict-free for the writers (one ray per lane) and the readers (8 consecutive rays per quarter).
__device__ __forceinline__ void tube_stage_operands(uint4 *stage, int lane, const uint32_t (&w)[2][8], bf16x8 (&A)[4])
{
    const int sw = (lane >> 1) & 3;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = q >> 1, kh = q & 1;
        stage[lane * 4 + (q ^ sw)] = make_uint4(w[k][4 * kh], w[k][4 * kh + 1], w[k][4 * kh + 2], w[k][4 * kh + 3]);
    }
    __builtin_amdgcn_wave_barrier();                // LDS ops of one wave execute in order
    const int row = lane & 31, kh = lane >> 5;
    const int q = 2 * ((row >> 3) & 1) + kh;
#pragma unroll
    for (int G = 0; G < 4; ++G) {
        const int ray = 16 * G + 8 * (row >> 4) + (row & 7);
        A[G] = __builtin_bit_cast(bf16x8, stage[ray * 4 + (q ^ ((ray >> 1) & 3))]);
    }
    __builtin_amdgcn_wave_barrier();
}

} // namespace rt

#ifdef RTIOW_CROSSCHECK_MODES
#include "xcheck/rt_xcheck_device.hpp"     // scan modes 3, 4: three-piece bf16 operands, the lifted form
#endif
