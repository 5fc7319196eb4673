// rt_xcheck_kernels.hpp -- part of the CROSS-CHECK build only (-DRTIOW_CROSSCHECK_MODES: tools/librtiow_hip_xcheck.so, a test artefact).
// Scan modes 2-4 are the earlier matrix-pipe forms of the sphere-scan filter; the product library carries modes 0, 1
// and 5 and never includes this file.  Known-answer kernels of modes 2-4 (rt_filter_products_device, rt_filter_lifted_device).
#pragma once
namespace rt {
// Known-answer hook for the matrix forms of the filter: one wave, 64 ray rows x 16 sphere
// columns; returns HB and Q exactly as the render kernel's tiles compute them
// (bf16x3 != 0: v_mfma_f32_16x16x32_bf16 on three-piece operands; else v_mfma_f32_16x16x4_f32).
__global__ __launch_bounds__(64) void filter_products_kernel(const float *r1, const float *r2, const float *s,
                                                           int bf16x3, float *hb_out, float *q_out)
{
    const int lane = threadIdx.x, col = lane & 15, quad = lane >> 4;
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    const float sv = s[col * 4 + quad];                     // S[k = quad][sphere = col]
    for (int G = 0; G < 4; ++G) {
        const float a1 = r1[(16 * G + col) * 4 + quad];     // R[ray 16G + (l&15)][k = l>>4]
        const float a2 = r2[(16 * G + col) * 4 + quad];
        f32x4 hb, q;
        if (bf16x3) {
            const Bf3 y = split_bf16x3(sv);
            const uint4 bw = make_uint4(y.p1 | (y.p2 << 16), y.p1 | (y.p3 << 16), y.p2 | (y.p1 << 16), y.p3 | (y.p2 << 16));
            const bf16x8 b = __builtin_bit_cast(bf16x8, bw);
            hb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_operand_bf16x3(a1), b, zero, 0, 0, 0);
            q = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_operand_bf16x3(a2), b, zero, 0, 0, 0);
        } else {
            hb = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, sv, zero, 0, 0, 0);
            q = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, sv, zero, 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) {                       // result (ray 16G + 4 quad + i, sphere col)
            hb_out[(16 * G + 4 * quad + i) * 16 + col] = hb[i];
            q_out[(16 * G + 4 * quad + i) * 16 + col] = q[i];
        }
    }
}

// One tile of the MODE 4 filter exactly as the render kernel evaluates it: 64 rays (o, d in f64,
// [64][3]) against the 16 columns of `tile` ([2][64] B operands built by the host exactly as
// rt_upload_scene builds them).  D_out[ray][column], R_out[ray][0..10] = the per-ray terms.
__global__ __launch_bounds__(64) void lifted_products_kernel(const double *o, const double *d, const uint4 *tile,
                                                           float *D_out, float *R_out)
{
    __shared__ uint4 stage[32 * kStageStride];
    const int lane = threadIdx.x, col = lane & 15, quad = lane >> 4;
    const LiftedRay L = make_lifted(mk(o[3 * lane], o[3 * lane + 1], o[3 * lane + 2]),
                                    mk(d[3 * lane], d[3 * lane + 1], d[3 * lane + 2]));
    for (int k = 0; k < kLiftTerms - 1; ++k) R_out[lane * kLiftTerms + k] = L.r[k];
    R_out[lane * kLiftTerms + kLiftTerms - 1] = L.sane ? 1.0f : 0.0f;
    bf16x8 A[4][2];
    uint32_t w[32];
    lifted_a_words(L, w);
    lifted_stage_operands(stage, lane, w, A);
    const bf16x8 b0 = __builtin_bit_cast(bf16x8, tile[lane]), b1 = __builtin_bit_cast(bf16x8, tile[64 + lane]);
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int G = 0; G < 4; ++G) {
        f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[G][0], b0, zero, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[G][1], b1, acc, 0, 0, 0);
        for (int i = 0; i < 4; ++i) D_out[(16 * G + 4 * quad + i) * 16 + col] = acc[i];
    }
}

} // namespace rt
