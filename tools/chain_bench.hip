// chain_bench.hip -- VALU issue rate of the scan's dependency shape: per "test" two
// 3-deep fma chains + 1 combining fma, with operands from VGPRs or SGPRs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s0, float s1, float s2, float s3)
{
    extern __shared__ unsigned char lds[];
    float gx = threadIdx.x * 1e-3f, gy = 0.5f, gz = 0.25f, h0 = 1.0f, px = 2.0f, py = 3.0f, pz = 4.0f, o2 = 5.0f;
    float acc = 0.0f;
    float cx = s0 + threadIdx.x, cy = s1, cz = s2;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float hb, q, D;
            if (KIND == 0) {          // all VGPR operands
                asm volatile("v_fma_f32 %0, %3, %6, %9\n v_fma_f32 %1, %3, %10, %13\n"
                             "v_fma_f32 %0, %4, %7, %0\n v_fma_f32 %1, %4, %11, %1\n"
                             "v_fma_f32 %0, %5, %8, %0\n v_fma_f32 %1, %5, %12, %1\n"
                             "v_fma_f32 %2, %0, %0, -%1\n"
                             : "=&v"(hb), "=&v"(q), "=&v"(D)
                             : "v"(cx), "v"(cy), "v"(cz), "v"(gx), "v"(gy), "v"(gz), "v"(h0), "v"(px), "v"(py), "v"(pz), "v"(o2));
            } else if (KIND == 1) {   // sphere values from SGPRs (as in the render kernel)
                asm volatile("v_fma_f32 %0, %3, %6, %9\n v_fma_f32 %1, %3, %10, %13\n"
                             "v_fma_f32 %0, %4, %7, %0\n v_fma_f32 %1, %4, %11, %1\n"
                             "v_fma_f32 %0, %5, %8, %0\n v_fma_f32 %1, %5, %12, %1\n"
                             "v_fma_f32 %2, %0, %0, -%1\n"
                             : "=&v"(hb), "=&v"(q), "=&v"(D)
                             : "s"(s0), "s"(s1), "s"(s2), "v"(gx), "v"(gy), "v"(gz), "v"(h0), "v"(px), "v"(py), "v"(pz), "v"(o2));
            } else if (KIND == 2) {   // SGPR operands with the negation modifier on the hb chain
                asm volatile("v_fma_f32 %0, -%3, %6, %9\n v_fma_f32 %1, %3, %10, %13\n"
                             "v_fma_f32 %0, -%4, %7, %0\n v_fma_f32 %1, %4, %11, %1\n"
                             "v_fma_f32 %0, -%5, %8, %0\n v_fma_f32 %1, %5, %12, %1\n"
                             "v_fma_f32 %2, %0, %0, -%1\n"
                             : "=&v"(hb), "=&v"(q), "=&v"(D)
                             : "s"(s0), "s"(s1), "s"(s2), "v"(gx), "v"(gy), "v"(gz), "v"(h0), "v"(px), "v"(py), "v"(pz), "v"(o2));
            } else if (KIND == 3) {   // two independent tests interleaved (ILP 4 chains), SGPR operands
                float hb2, q2, D2;
                asm volatile("v_fma_f32 %0, %6, %9, %12\n v_fma_f32 %1, %6, %13, %16\n v_fma_f32 %3, %17, %9, %12\n v_fma_f32 %4, %17, %13, %16\n"
                             "v_fma_f32 %0, %7, %10, %0\n v_fma_f32 %1, %7, %14, %1\n v_fma_f32 %3, %7, %10, %3\n v_fma_f32 %4, %7, %14, %4\n"
                             "v_fma_f32 %0, %8, %11, %0\n v_fma_f32 %1, %8, %15, %1\n v_fma_f32 %3, %8, %11, %3\n v_fma_f32 %4, %8, %15, %4\n"
                             "v_fma_f32 %2, %0, %0, -%1\n v_fma_f32 %5, %3, %3, -%4\n"
                             : "=&v"(hb), "=&v"(q), "=&v"(D), "=&v"(hb2), "=&v"(q2), "=&v"(D2)
                             : "s"(s0), "s"(s1), "s"(s2), "v"(gx), "v"(gy), "v"(gz), "v"(h0), "v"(px), "v"(py), "v"(pz), "v"(o2), "s"(s3));
                acc += D2;
            } else if (KIND == 4) {   // fmac (VOP2, 4-byte encoding) where possible
                asm volatile("v_fma_f32 %0, %3, %6, %9\n v_fma_f32 %1, %3, %10, %13\n"
                             "v_fmac_f32 %0, %4, %7\n v_fmac_f32 %1, %4, %11\n"
                             "v_fmac_f32 %0, %5, %8\n v_fmac_f32 %1, %5, %12\n"
                             "v_fma_f32 %2, %0, %0, -%1\n"
                             : "=&v"(hb), "=&v"(q), "=&v"(D)
                             : "s"(s0), "s"(s1), "s"(s2), "v"(gx), "v"(gy), "v"(gz), "v"(h0), "v"(px), "v"(py), "v"(pz), "v"(o2));
            }
            acc += D;       // one more dependent VALU per test (8 per test in total)
            gx += 1e-7f;    // and one independent (9 per test)
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int KIND>
void run(const char *name, int cus, float *d_out, int valu_per_test)
{
    const int iters = 40000;
    for (int bpc : {2, 4, 5, 8}) {
        size_t lds = (160 * 1024 / bpc) & ~255;
        CHECK(hipFuncSetAttribute((const void *)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        int grid = cus * bpc;
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), lds, 0, d_out, 2000, 1.0f, 2.0f, 3.0f, 4.0f);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), lds, 0, d_out, iters, 1.0f, 2.0f, 3.0f, 4.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        double tests_per_simd = (double)bpc * iters * 8 * (KIND == 3 ? 2 : 1);
        double cyc = ms * 1e-3 * 2.4e9 / tests_per_simd;
        printf("%-34s waves/SIMD=%d %8.3f ms  %.2f cyc/wave-test  %.2f cyc/VALU @2.4GHz\n", name, bpc, ms, cyc, cyc / valu_per_test);
    }
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    float *d_out; CHECK(hipMalloc(&d_out, 256 * 8 * p.multiProcessorCount * 4));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<0>, dim3(p.multiProcessorCount * 8), dim3(256), 8192, 0, d_out, 40000, 1.f, 2.f, 3.f, 4.f);
    CHECK(hipDeviceSynchronize());
    run<0>("vgpr operands", p.multiProcessorCount, d_out, 9);
    run<1>("sgpr operands", p.multiProcessorCount, d_out, 9);
    run<2>("sgpr operands, neg modifier", p.multiProcessorCount, d_out, 9);
    run<4>("sgpr operands, fmac where possible", p.multiProcessorCount, d_out, 9);
    run<3>("sgpr, two tests interleaved", p.multiProcessorCount, d_out, 9);
    return 0;
}
