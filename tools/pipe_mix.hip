// pipe_mix.hip -- do f64 and f32 vector instructions of DIFFERENT waves of one SIMD share an issue pipe?  512-thread workgroups (waves w and
// w + 4 share a SIMD), two per CU = 4 waves per SIMD; each wave runs N instructions of one kind, chosen by (wave >> 2) & 1.
// all f32 / all f64 / half and half: if the mixed run takes the SUM of the halves the pipe is shared, if the MAX the pipes are separate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define R8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
template <int KA, int KB>
__global__ __launch_bounds__(512, 2) void k(int iters, float *out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float x[8]; double d[8];
    for (int i = 0; i < 8; ++i) { x[i] = (float)(lane + i); d[i] = (double)(lane + i) * 1.0000001; }
    float y = 0.999f; double dy = 1.0000001, dz = 1e-9;
    unsigned long long acc = lane;
    const int kind = ((wave >> 2) & 1) ? KB : KA;
#define SF(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(y));
#define SX(i) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(x[i]) : "v"(y));
#define SD(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dy), "v"(dz));
#define SM(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(d[i]) : "v"(y) : "vcc");
#define SC(i) asm volatile("v_cmp_lt_f64_e32 vcc, %0, %1" : : "v"(d[i]), "v"(dy) : "vcc");
    for (int it = 0; it < iters; ++it) {
        if (kind == 0) { R8(SF) R8(SF) } else if (kind == 1) { R8(SD) R8(SD) } else if (kind == 2) { R8(SX) R8(SX) }
        else if (kind == 3) { R8(SM) R8(SM) } else if (kind == 4) { R8(SC) R8(SC) }
        else { }   // kind 5: idle wave
    }
    float r = (float)acc;
    for (int i = 0; i < 8; ++i) r += x[i] + (float)d[i];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}
template <int KA, int KB> float run(int cus, float *d_out)
{
    const int iters = 20000;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<KA, KB>), dim3(cus * 2), dim3(512), 0, 0, 100, d_out);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<KA, KB>), dim3(cus * 2), dim3(512), 0, 0, iters, d_out);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return (float)(ms * 1e-3 * 2.3e9 / ((double)iters * 16));       // SIMD cycles (at 2.3 GHz) per instruction-slot of ONE wave
}
int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *d_out; CHECK(hipMalloc(&d_out, cus * 2 * 512 * 4));
    const char *names[] = {"v_fma_f32", "v_fma_f64", "v_xor_b32", "v_mad_u64_u32", "v_cmp_lt_f64", "idle"};
    printf("4 waves per SIMD; waves 0-3 of a workgroup run kind A, waves 4-7 kind B (w and w+4 share a SIMD); cycles per instruction of one wave's stream\n");
#define ROW(A, B) printf("  A = %-14s B = %-14s  %7.2f\n", names[A], names[B], run<A, B>(cus, d_out));
    ROW(0, 0) ROW(1, 1) ROW(0, 1) ROW(0, 5) ROW(1, 5)
    ROW(2, 2) ROW(2, 1) ROW(2, 5)
    ROW(3, 3) ROW(3, 1) ROW(3, 2) ROW(3, 5)
    ROW(4, 4) ROW(4, 2) ROW(4, 1)
    return 0;
}
