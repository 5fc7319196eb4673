// exec_mask_cost.hip -- does a vector instruction cost less when part of the wave is masked off?  SIMD cycles per
// instruction at 4 waves per SIMD for a stream of ONE kind of instruction executed under an EXEC mask of 64, 32 (low half),
// 32 (even lanes), 16, 8 and 1 active lanes.  (The render kernel's divergent phases -- retry loops, partial exact-test
// rounds, material branches -- run with few active lanes: is that time proportional to instructions or to active lanes?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define BODY16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

template <int KIND>
__global__ __launch_bounds__(256, 4) void k(int iters, float *out, unsigned long long mask)
{
    const int lane = threadIdx.x & 63;
    float x[8];
    double d[8];
    for (int i = 0; i < 8; ++i) { x[i] = (float)(lane + i); d[i] = (double)(lane + i) * 1.0000001; }
    float y = -(float)lane, z = 0.5f;
    double dy = 1.0000001, dz = 1e-9;
    if ((mask >> lane) & 1ull) {
        for (int it = 0; it < iters; ++it) {
#define S_ADD32(i) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(x[i]) : "v"(y));
#define S_BITOP3(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(x[i]) : "v"(y), "v"(z));
#define S_MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "+v"(d[i]) : "v"(y), "v"(z) : "vcc");
#define S_MADI64(i) asm volatile("v_mad_i64_i32 %0, vcc, %1, %1, %0" : "+v"(d[i]) : "v"(y) : "vcc");
#define S_ADDF64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dz));
#define S_MULF64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dy));
#define S_FMAF64(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dy), "v"(dz));
#define S_CMP(i) asm volatile("v_cmp_ge_f32_e32 vcc, %1, %0" : : "v"(x[i]), "v"(y) : "vcc");
            if (KIND == 0) { BODY16(S_ADD32) } else if (KIND == 1) { BODY16(S_BITOP3) } else if (KIND == 2) { BODY16(S_MAD64) }
            else if (KIND == 3) { BODY16(S_MADI64) } else if (KIND == 4) { BODY16(S_ADDF64) } else if (KIND == 5) { BODY16(S_MULF64) }
            else if (KIND == 6) { BODY16(S_FMAF64) } else if (KIND == 7) { BODY16(S_CMP) }
        }
    }
    float r = 0.0f;
    for (int i = 0; i < 8; ++i) r += x[i] + (float)d[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int KIND>
void run(const char *name, int cus, float *d_out)
{
    const int iters = 20000;
    const unsigned long long masks[12] = {~0ull, 0xFFFFFFFFull, 0x5555555555555555ull, 0xFFFFull, 0xFFFull, 0x1FFull, 0xFFull, 0xFF00ull,
                                          0x0101010101010101ull, 0x1111111111111111ull, 0x8000000000000000ull, 1ull};
    printf("%-18s", name);
    for (unsigned long long m : masks) {
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<KIND>, dim3(cus * 4), dim3(256), 0, 0, 100, d_out, m);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(cus * 4), dim3(256), 0, 0, iters, d_out, m);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf(" %6.2f", (float)(ms * 1e-3 * 2.3e9 / (4.0 * iters * 16)));
    }
    printf("\n");
}

int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *d_out; CHECK(hipMalloc(&d_out, cus * 4 * 256 * 4));
    printf("SIMD cycles per instruction at 4 waves/SIMD; active lanes: 64 | 32 low half | 32 even lanes | 16 (0-15) | 12 (0-11) | 9 (0-8) | 8 (0-7) | 8 (8-15) | 8 spread (every 8th) | 16 spread (every 4th) | lane 63 | lane 0\n");
    run<0>("v_add_f32_e32", cus, d_out);
    run<1>("v_bitop3_b32", cus, d_out);
    run<7>("v_cmp_ge_f32_e32", cus, d_out);
    run<2>("v_mad_u64_u32", cus, d_out);
    run<3>("v_mad_i64_i32", cus, d_out);
    run<4>("v_add_f64", cus, d_out);
    run<5>("v_mul_f64", cus, d_out);
    run<6>("v_fma_f64", cus, d_out);
    return 0;
}
