// mfma_bench.hip -- issue rate of v_mfma_f32_16x16x4_f32 (VGPR destination), alone and with
// the filter's VALU epilogue interleaved (4 fma + max per MFMA pair).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s)
{
    extern __shared__ unsigned char lds[];
    const float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    float b = s + threadIdx.x;
    const f32x4 z = {0, 0, 0, 0};
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        f32x4 r0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, z, 0, 0, 0);
        f32x4 r1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, z, 0, 0, 0);
        f32x4 r2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b, z, 0, 0, 0);
        f32x4 r3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b, z, 0, 0, 0);
        f32x4 r4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b + 1.0f, z, 0, 0, 0);
        f32x4 r5 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b + 1.0f, z, 0, 0, 0);
        f32x4 r6 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b + 1.0f, z, 0, 0, 0);
        f32x4 r7 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b + 1.0f, z, 0, 0, 0);
        if (KIND == 0) {
            acc += r0[0] + r1[0] + r2[0] + r3[0] + r4[0] + r5[0] + r6[0] + r7[0];
        } else {
            float m = -1e30f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                m = fmaxf(m, fmaf(r0[i], r0[i], -r4[i]));
                m = fmaxf(m, fmaf(r1[i], r1[i], -r5[i]));
                m = fmaxf(m, fmaf(r2[i], r2[i], -r6[i]));
                m = fmaxf(m, fmaf(r3[i], r3[i], -r7[i]));
            }
            acc += m;
        }
        b += 1e-6f;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int KIND>
void run(const char *name, int cus, float *d_out)
{
    const int iters = 100000;
    for (int bpc : {1, 2, 3, 4}) {
        size_t lds = (160 * 1024 / bpc) & ~255;
        CHECK(hipFuncSetAttribute((const void *)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        int grid = cus * bpc;
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), lds, 0, d_out, 5000, 1.0f);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), lds, 0, d_out, iters, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        double mfma_per_simd = (double)bpc * iters * 8;
        printf("%-28s waves/SIMD=%d %8.3f ms  %.2f cycles/MFMA/SIMD @2.4GHz\n", name, bpc, ms, ms * 1e-3 * 2.4e9 / mfma_per_simd);
    }
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    float *d_out; CHECK(hipMalloc(&d_out, 256 * 8 * p.multiProcessorCount * 4));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<0>, dim3(p.multiProcessorCount * 4), dim3(256), 8192, 0, d_out, 100000, 1.f);
    CHECK(hipDeviceSynchronize());
    run<0>("mfma 16x16x4 f32 only", p.multiProcessorCount, d_out);
    run<1>("mfma + filter epilogue", p.multiProcessorCount, d_out);
    return 0;
}
