// mfma_rate.hip -- issue rate of the bf16 MFMA shapes on gfx950 (cycles per instruction per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void k(int iters, float *out, unsigned long long *cyc)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_readcyclecounter();
    const int lane = threadIdx.x & 63;
    uint4 w = make_uint4(0x3c003c00u + lane, 0x3c103c10u, 0xbc00bc00u, 0x3c003c00u);
    const bf16x8 a8 = __builtin_bit_cast(bf16x8, w);
    const s16x4 a4 = {(short)(0x3c00 + lane), 0x3c10, (short)0xbc00, 0x3c00};
    float res = 0.0f;
    if (KIND == 0) {
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, a8, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, a8, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, a8, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, a8, c3, 0, 0, 0);
        }
        res = c0[0] + c1[1] + c2[2] + c3[3];
    } else if (KIND == 1) {
        f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, a8, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, a8, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, a8, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, a8, c3, 0, 0, 0);
        }
        res = c0[0] + c1[1] + c2[2] + c3[3];
    } else if (KIND == 2) {
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, a4, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, a4, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, a4, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, a4, c3, 0, 0, 0);
        }
        res = c0[0] + c1[1] + c2[2] + c3[3];
    } else {
        f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, a4, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, a4, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, a4, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, a4, c3, 0, 0, 0);
        }
        res = c0[0] + c1[1] + c2[2] + c3[3];
    }
    out[blockIdx.x * 256 + threadIdx.x] = res;
    if (blockIdx.x == 0 && threadIdx.x == 0) { __builtin_amdgcn_s_waitcnt(0); cyc[0] = __builtin_amdgcn_s_memtime() - t0; cyc[1] = __builtin_readcyclecounter() - r0; }
}

template <int KIND>
void run(const char *name, int cus, float *d_out)
{
    unsigned long long *d_cyc; CHECK(hipMalloc(&d_cyc, 16)); unsigned long long h_cyc[2];
    const int iters = 20000;
    for (int bpc : {1, 2, 4}) {
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<KIND>, dim3(cus * bpc), dim3(256), 0, 0, 100, d_out, d_cyc);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(cus * bpc), dim3(256), 0, 0, iters, d_out, d_cyc);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(h_cyc, d_cyc, 16, hipMemcpyDeviceToHost));
        printf("%-28s waves/SIMD=%d %8.3f ms  %6.2f ns per MFMA per SIMD; wave 0: memtime %llu ticks, cyclecounter %llu -> %.2f / %.2f per MFMA per SIMD\n", name, bpc, ms, ms * 1e6 / ((double)bpc * iters * 4), h_cyc[0], h_cyc[1], (double)h_cyc[0] / (bpc * iters * 4.0), (double)h_cyc[1] / (bpc * iters * 4.0));
    }
}

int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *d_out; CHECK(hipMalloc(&d_out, cus * 4 * 256 * 4));
    run<0>("v_mfma_f32_16x16x32_bf16", cus, d_out);
    run<1>("v_mfma_f32_32x32x16_bf16", cus, d_out);
    run<2>("v_mfma_f32_16x16x16_bf16 (1k)", cus, d_out);
    run<3>("v_mfma_f32_32x32x8_bf16 (1k)", cus, d_out);
    return 0;
}
