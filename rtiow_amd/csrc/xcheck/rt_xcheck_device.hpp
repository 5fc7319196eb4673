// rt_xcheck_device.hpp -- part of the CROSS-CHECK build only (-DRTIOW_CROSSCHECK_MODES: tools/librtiow_hip_xcheck.so, a test artefact).
// Device-side arithmetic of scan modes 3 and 4, the earlier bf16 matrix-pipe forms of the sphere-scan filter: the
// three-piece bf16 split of an f32 operand and the "lifted" single-contraction form.  The product library (modes 0, 1, 5) never includes it.
#pragma once
namespace rt {

// ---- bf16x3: an f32 value as the exact sum of three bf16 pieces ------------------------
// x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1), x3 = x - x1 - x2 (8 significant bits
// each, round to nearest even; the remainders are exact in f32).  A product x*y then is the
// sum of nine exact bf16 x bf16 products; the matrix form keeps the eight of relative size
// >= 2^-24 and drops x3*y3 (<= 2^-32).
__device__ __forceinline__ uint32_t bf16_rne_bits(float x)
{
    const uint32_t u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
struct Bf3 { uint32_t p1, p2, p3; };    // bf16 bit patterns in the low 16 bits
__device__ __forceinline__ Bf3 split_bf16x3(float x)
{
    Bf3 r;
    r.p1 = bf16_rne_bits(x);
    const float r1 = x - __uint_as_float(r.p1 << 16);
    r.p2 = bf16_rne_bits(r1);
    const float r2 = r1 - __uint_as_float(r.p2 << 16);
    r.p3 = bf16_rne_bits(r2);
    return r;
}
// A-side element order (x1,x1,x2,x1,x2,x3,x2,x3) against the B-side order
// (y1,y2,y1,y3,y2,y1,y3,y2) built on the host: the 8 products listed above.
__device__ __forceinline__ bf16x8 a_operand_bf16x3(float x)
{
    const Bf3 s = split_bf16x3(x);
    uint4 w;
    w.x = s.p1 | (s.p1 << 16);      // x1, x1
    w.y = s.p2 | (s.p1 << 16);      // x2, x1
    w.z = s.p2 | (s.p3 << 16);      // x2, x3
    w.w = s.p2 | (s.p3 << 16);      // x2, x3
    return __builtin_bit_cast(bf16x8, w);
}

// ---- MODE 4: the filter as ONE contraction ("lifted" form) ------------------------------
// With g = d / sqrt(a (1-KU)) (so |g| >= 1) and kappa = KU / (1-KU),
//
//   (g.(o-c))^2 - |o-c|^2 + r^2 + kappa (|o|^2 + |c|^2 + 2 r^2)  =  sum_k R_k(o,d) * C_k(c,r)
//
//   R = ( (g.o)^2 - |o|^2 (1-kappa),   2 (o_j - (g.o) g_j)  j=x,y,z,
//         gx^2, gy^2, gz^2, 2 gx gy, 2 gx gz, 2 gy gz,   1 )
//   C = ( 1,   c_j,   cx^2, cy^2, cz^2, cx cy, cx cz, cy cz,   r^2 (1+2 kappa) - |c|^2 (1-kappa) )
//
// The left side is >= disc/a + kappa S (S = |o|^2 + |c|^2 + r^2), so "sum >= 0" keeps every sphere
// the reference can hit as long as the evaluation error stays below kappa S.  The
// eleven products run on the bf16 matrix pipe as 64 piece-products (two chained K = 32 MFMAs):
// each of the nine general products x*y contributes the six pieces (x1y1, x1y2, x2y1, x1y3, x2y2,
// x3y1) -- what is dropped is <= 2.01 * 2^-24 |x y| -- and the two products with a factor 1
// contribute three.  The VALU only looks at the SIGN of the result.
constexpr float kFilterKU_lifted = 1024.0f * kUnitRoundoff;             // 2^-14
constexpr int kLiftTerms = 11;
constexpr uint32_t kBf16One = 0x3F80u;
constexpr float kLiftNever = -3.0e38f;      // C_10 of a column that must never be kept (padding, always-exact)
constexpr float kLiftAlways = 3.0e38f;      // C_10 of a column outside the analysed range: always kept

struct LiftedRay {
    float r[kLiftTerms - 1];    // R_0 .. R_9 (R_10 = 1 is implicit)
    bool sane;                  // false: outside the analysed range, every sphere must be tested exactly
};

__device__ __forceinline__ LiftedRay make_lifted(D3 o, D3 d)
{
    constexpr float KU = kFilterKU_lifted;
    LiftedRay L;
    const float ox = (float)o.x, oy = (float)o.y, oz = (float)o.z;
    const float dx = (float)d.x, dy = (float)d.y, dz = (float)d.z;
    const float a = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    const float oo = __builtin_fmaf(oz, oz, __builtin_fmaf(oy, oy, ox * ox));
    const float s = __builtin_amdgcn_rsqf(a * (1.0f - KU));                 // v_rsq_f32, 1 ulp
    const float gx = dx * s, gy = dy * s, gz = dz * s;
    const float go = __builtin_fmaf(oz, gz, __builtin_fmaf(oy, gy, ox * gx));
    L.r[0] = __builtin_fmaf(go, go, -(oo * (1.0f - KU / (1.0f - KU))));
    L.r[1] = 2.0f * __builtin_fmaf(-go, gx, ox);
    L.r[2] = 2.0f * __builtin_fmaf(-go, gy, oy);
    L.r[3] = 2.0f * __builtin_fmaf(-go, gz, oz);
    L.r[4] = gx * gx; L.r[5] = gy * gy; L.r[6] = gz * gz;
    L.r[7] = 2.0f * (gx * gy); L.r[8] = 2.0f * (gx * gz); L.r[9] = 2.0f * (gy * gz);
    L.sane = (a > 1e-20f) && (a < 1e20f) && (oo < 1e30f);
    if (!L.sane) {
#pragma unroll
        for (int k = 0; k < kLiftTerms - 1; ++k) L.r[k] = 0.0f;
    }
    return L;
}
// a lane without a ray: the sum is hugely negative for every real column
__device__ __forceinline__ LiftedRay no_lifted_ray()
{
    LiftedRay L;
    L.r[0] = -1e30f;
#pragma unroll
    for (int k = 1; k < kLiftTerms - 1; ++k) L.r[k] = 0.0f;
    L.sane = true;
    return L;
}

// the same three-piece split with the hardware's RNE conversion (v_cvt_pk_bf16_f32)
__device__ __forceinline__ Bf3 split_bf16x3_hw(float x)
{
    Bf3 r;
    r.p1 = (uint32_t)__builtin_bit_cast(unsigned short, (__bf16)x);
    const float r1 = x - __uint_as_float(r.p1 << 16);
    r.p2 = (uint32_t)__builtin_bit_cast(unsigned short, (__bf16)r1);
    const float r2 = r1 - __uint_as_float(r.p2 << 16);
    r.p3 = __float_as_uint(r2) >> 16;           // at most 8 significant bits are left: exact
    return r;
}

// The 64 K-slots of the contraction as 32 dwords (two bf16 each, low half = even slot):
//   slots 6p .. 6p+5  (p = 0..8, term 1+p):  A (x1,x1,x2,x1,x2,x3)   B (y1,y2,y1,y3,y2,y1)
//   slots 54,55,56    (term 0):              A (R0_1,R0_2,R0_3)      B (1,1,1)
//   slots 57,58,59    (term 10):             A (1,1,1)               B (C10_1,C10_2,C10_3)
//   slots 60..63:                            zero
__device__ __forceinline__ void lifted_a_words(const LiftedRay &L, uint32_t (&w)[32])
{
#pragma unroll
    for (int p = 0; p < 9; ++p) {
        const Bf3 x = split_bf16x3_hw(L.r[1 + p]);
        w[3 * p + 0] = x.p1 | (x.p1 << 16);
        w[3 * p + 1] = x.p2 | (x.p1 << 16);
        w[3 * p + 2] = x.p2 | (x.p3 << 16);
    }
    const Bf3 x0 = split_bf16x3_hw(L.r[0]);
    w[27] = x0.p1 | (x0.p2 << 16);
    w[28] = x0.p3 | (kBf16One << 16);
    w[29] = kBf16One | (kBf16One << 16);
    w[30] = 0u; w[31] = 0u;
}

// Per-ray dwords -> MFMA A operands through LDS.  For v_mfma_f32_16x16x32_bf16 lane l holds row
// (ray) l&15 and the K-slots 8(l>>4) .. 8(l>>4)+7; operand A[G][m] serves rays 16G..16G+15 and
// slots 32m..32m+31.  `stage` is 32*kStageStride uint4 of LDS owned by this wave: the 64 rays go
// through in two rounds of 32 (row stride 9 uint4: conflict-free 16-byte accesses).
__device__ __forceinline__ void lifted_stage_operands(uint4 *stage, int lane, const uint32_t (&w)[32], bf16x8 (&A)[4][2])
{
    const int col = lane & 15, quad = lane >> 4;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if ((lane >> 5) == half) {
            uint4 *row = stage + (lane & 31) * kStageStride;
#pragma unroll
            for (int k = 0; k < 8; ++k) row[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
        }
        __builtin_amdgcn_wave_barrier();            // LDS ops of one wave execute in order
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int m = 0; m < 2; ++m)
                A[2 * half + g][m] = __builtin_bit_cast(bf16x8, stage[(16 * g + col) * kStageStride + 4 * m + quad]);
        __builtin_amdgcn_wave_barrier();
    }
}

} // namespace rt
