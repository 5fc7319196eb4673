// mfma_valu_mix.hip -- every wave runs the SAME stream: one v_mfma_f32_32x32x16_bf16 followed by N independent
// vector instructions on registers the MFMA neither reads nor writes (no hazards, no wait states).  SIMD cycles per
// (MFMA + N VALU) unit at 1, 2, 4 waves per SIMD, for N = 0..28 and for three kinds of vector instruction.
// Question: does the matrix pipe's 32 cycles per MFMA overlap with the vector instructions of the OTHER waves of the
// SIMD when all waves interleave both?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// KIND 0: v_max_f32 with |.| modifiers (VOP3)   1: v_min3_i32   2: v_add_f32 (VOP2)   3: v_bitop3_b32   4: v_or_b32
// MF: issue the MFMA or not.  AG: the MFMA's A and B operands live in AGPRs (the accumulator half of the register file)
template <int N, int KIND, bool MF, bool AG = false>
__global__ __launch_bounds__(256, 4) void k(int iters, float *out)
{
    const int lane = threadIdx.x & 63;
    uint4 w = make_uint4(0x3c003c00u + lane, 0x3c103c10u, 0xbc00bc00u, 0x3c003c00u);
    const bf16x8 a = __builtin_bit_cast(bf16x8, w), b = a;
    f32x16 acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc1 = acc0;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = (float)(lane + i);
    float y = -(float)lane, z = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (MF && AG) {
                if (u == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc0) : "a"(a), "a"(b));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc1) : "a"(a), "a"(b));
            } else if (MF) {
                if (u == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            }
#pragma unroll
            for (int n = 0; n < N; ++n) {
                if (KIND == 0) asm volatile("v_max_f32_e64 %0, |%0|, |%1|" : "+v"(x[n & 7]) : "v"(y));
                else if (KIND == 1) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(x[n & 7]) : "v"(y), "v"(z));
                else if (KIND == 2) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(x[n & 7]) : "v"(y));
                else if (KIND == 3) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe0" : "+v"(x[n & 7]) : "v"(y), "v"(z));
                else asm volatile("v_or_b32_e32 %0, %1, %0" : "+v"(x[n & 7]) : "v"(y));
            }
        }
    }
    float r = acc0[0] + acc1[3];
    for (int i = 0; i < 8; ++i) r += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int N, int KIND, bool MF, bool AG = false>
float run(int cus, int bpc, float *d_out)
{
    const int iters = 20000;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<N, KIND, MF, AG>), dim3(cus * bpc), dim3(256), 0, 0, 100, d_out);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<N, KIND, MF, AG>), dim3(cus * bpc), dim3(256), 0, 0, iters, d_out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return (float)(ms * 1e-3 * 2.3e9 / ((double)bpc * iters * 2));      // SIMD cycles per unit at 2.3 GHz
}

template <int N, int KIND>
void row(const char *kind, int cus, float *d_out)
{
    printf("%-12s N=%2d   with MFMA: %6.1f %6.1f %6.1f   VALU only: %6.1f %6.1f %6.1f   (cycles per unit at 1, 2, 4 waves/SIMD)\n", kind, N,
           run<N, KIND, true>(cus, 1, d_out), run<N, KIND, true>(cus, 2, d_out), run<N, KIND, true>(cus, 4, d_out),
           run<N, KIND, false>(cus, 1, d_out), run<N, KIND, false>(cus, 2, d_out), run<N, KIND, false>(cus, 4, d_out));
}

int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *d_out; CHECK(hipMalloc(&d_out, cus * 4 * 256 * 4));
    row<0, 0>("v_max|f32|", cus, d_out);
    row<4, 0>("v_max|f32|", cus, d_out);
    row<8, 0>("v_max|f32|", cus, d_out);
    row<14, 0>("v_max|f32|", cus, d_out);
    row<20, 0>("v_max|f32|", cus, d_out);
    row<28, 0>("v_max|f32|", cus, d_out);
    row<14, 1>("v_min3_i32", cus, d_out);
    row<28, 1>("v_min3_i32", cus, d_out);
    row<14, 2>("v_add_f32", cus, d_out);
    row<28, 2>("v_add_f32", cus, d_out);
    row<9, 3>("v_bitop3_b32", cus, d_out);
    row<18, 3>("v_bitop3_b32", cus, d_out);
    row<9, 4>("v_or_b32", cus, d_out);
    printf("-- MFMA with A and B in AGPRs: cycles per unit at 1, 2, 4 waves/SIMD --\n");
    printf("v_add_f32    N=14   %6.1f %6.1f %6.1f\n", run<14, 2, true, true>(cus, 1, d_out), run<14, 2, true, true>(cus, 2, d_out), run<14, 2, true, true>(cus, 4, d_out));
    printf("v_bitop3_b32 N= 9   %6.1f %6.1f %6.1f\n", run<9, 3, true, true>(cus, 1, d_out), run<9, 3, true, true>(cus, 2, d_out), run<9, 3, true, true>(cus, 4, d_out));
    printf("v_max|f32|   N=14   %6.1f %6.1f %6.1f\n", run<14, 0, true, true>(cus, 1, d_out), run<14, 0, true, true>(cus, 2, d_out), run<14, 0, true, true>(cus, 4, d_out));
    return 0;
}
