// overlap_bench.hip -- do bf16 MFMA waves and VALU waves on the SAME SIMD overlap?
// Blocks of 512 threads: waves 0-3 (one per SIMD) run an MFMA loop, waves 4-7 a VALU loop
// (f32 fma, f64 fma, int mad, v_max_f32 with |.| modifiers, v_min3_i32).  Time(MFMA alone), time(VALU alone),
// time(both).  MFMA shape: 16x16x32 (SHAPE 0) or the render kernel's 32x32x16 (SHAPE 1).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int VKIND, int SHAPE>
__global__ __launch_bounds__(512) void k(int mfma_iters, int valu_iters, float *out)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float res = 0.0f;
    if (wave < 4) {
        uint4 w = make_uint4(0x3c003c00u + lane, 0x3c103c10u, 0xbc00bc00u, 0x3c003c00u);
        const bf16x8 a = __builtin_bit_cast(bf16x8, w);
        bf16x8 b = a;
        if (SHAPE == 0) {
            f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
            for (int i = 0; i < mfma_iters; ++i) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
            }
            res = c0[0] + c1[1] + c2[2] + c3[3];
        } else {
            f32x16 c0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, c1 = c0;
            for (int i = 0; i < mfma_iters; ++i) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            }
            res = c0[0] + c1[1];
        }
    } else {
        if (VKIND == 0) {
            float x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3, m = 1.0001f;
            for (int i = 0; i < valu_iters; ++i) {
                x0 = __builtin_fmaf(x0, m, 1.0f); x1 = __builtin_fmaf(x1, m, 1.0f);
                x2 = __builtin_fmaf(x2, m, 1.0f); x3 = __builtin_fmaf(x3, m, 1.0f);
            }
            res = x0 + x1 + x2 + x3;
        } else if (VKIND == 1) {
            double x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3, m = 1.0001;
            for (int i = 0; i < valu_iters; ++i) {
                x0 = __builtin_fma(x0, m, 1.0); x1 = __builtin_fma(x1, m, 1.0);
                x2 = __builtin_fma(x2, m, 1.0); x3 = __builtin_fma(x3, m, 1.0);
            }
            res = (float)(x0 + x1 + x2 + x3);
        } else if (VKIND == 2) {
            unsigned x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3;
            for (int i = 0; i < valu_iters; ++i) {
                x0 = x0 * 0x9E3779B9u + 7u; x1 = x1 * 0x9E3779B9u + 7u;
                x2 = x2 * 0x9E3779B9u + 7u; x3 = x3 * 0x9E3779B9u + 7u;
            }
            res = (float)(x0 ^ x1 ^ x2 ^ x3);
        } else if (VKIND == 3) {                    // the look's v_max_f32 with magnitude modifiers
            float x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3, y = -(float)lane;
            for (int i = 0; i < valu_iters; ++i) {
                asm volatile("v_max_f32_e64 %0, |%0|, |%4|\n v_max_f32_e64 %1, |%1|, |%4|\n v_max_f32_e64 %2, |%2|, |%4|\n v_max_f32_e64 %3, |%3|, |%4|"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y));
            }
            res = x0 + x1 + x2 + x3;
        } else {                                    // the look's v_min3_i32
            int x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3, y = 1000 - lane, z = 77;
            for (int i = 0; i < valu_iters; ++i) {
                asm volatile("v_min3_i32 %0, %0, %4, %5\n v_min3_i32 %1, %1, %4, %5\n v_min3_i32 %2, %2, %4, %5\n v_min3_i32 %3, %3, %4, %5"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y), "v"(z));
            }
            res = (float)(x0 + x1 + x2 + x3);
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = res;
}

template <int VKIND, int SHAPE>
void run(const char *name, int cus, float *d_out)
{
    for (int bpc : {1, 2}) {
        float t[3];
        const int mi = SHAPE == 0 ? 20000 : 10000;
        const int cfg[3][2] = {{mi, 0}, {0, 80000}, {mi, 80000}};
        for (int c = 0; c < 3; ++c) {
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            hipLaunchKernelGGL((k<VKIND, SHAPE>), dim3(cus * bpc), dim3(512), 0, 0, 100, 100, d_out);
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL((k<VKIND, SHAPE>), dim3(cus * bpc), dim3(512), 0, 0, cfg[c][0], cfg[c][1], d_out);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&t[c], e0, e1));
        }
        printf("%-10s blocks/CU=%d (%d MFMA + %d VALU waves per SIMD): MFMA alone %.3f ms, VALU alone %.3f ms, both %.3f ms  (max %.3f, sum %.3f)\n",
               name, bpc, bpc, bpc, t[0], t[1], t[2], t[0] > t[1] ? t[0] : t[1], t[0] + t[1]);
    }
}

int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *d_out; CHECK(hipMalloc(&d_out, cus * 2 * 512 * 4));
    run<0, 0>("f32 fma", cus, d_out);
    run<1, 0>("f64 fma", cus, d_out);
    run<2, 0>("u32 mad", cus, d_out);
    printf("-- 32x32x16 MFMA --\n");
    run<0, 1>("f32 fma", cus, d_out);
    run<3, 1>("v_max|f32|", cus, d_out);
    run<4, 1>("v_min3_i32", cus, d_out);
    run<2, 1>("u32 mad", cus, d_out);
    return 0;
}
