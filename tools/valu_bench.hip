// valu_bench.hip -- measures per-instruction VALU issue rates on gfx950 to size
// the render kernel's roofline.  Each kernel runs a long
// chain-free stream of one instruction kind in every lane; the host sweeps the
// number of waves per SIMD.  Output: one line per (instruction, waves/SIMD):
// cycles per wave-instruction per SIMD, derived from wall time and s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 16;   // independent accumulators

template <int KIND>
__global__ __launch_bounds__(256) void bench(float *out, unsigned long long *cyc, float seed)
{
    float a[UNROLL];
    double da[UNROLL / 2];
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f pa[UNROLL / 2];
    unsigned long long ua[UNROLL / 2];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) a[k] = seed + k + threadIdx.x;
#pragma unroll
    for (int k = 0; k < UNROLL / 2; ++k) { da[k] = seed + k + threadIdx.x; pa[k] = v2f{a[2 * k], a[2 * k + 1]}; ua[k] = threadIdx.x + k; }
    const float m = seed * 0.5f, c = seed * 0.25f;
    const double dm = m, dc = c;
    const v2f pm = v2f{m, m}, pc = v2f{c, c};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
        if (KIND == 0) {
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
        } else if (KIND == 1) {
#pragma unroll
            for (int k = 0; k < UNROLL / 2; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pa[k]) : "v"(pm), "v"(pc));
        } else if (KIND == 2) {
#pragma unroll
            for (int k = 0; k < UNROLL / 2; ++k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(da[k]) : "v"(dm), "v"(dc));
        } else if (KIND == 3) {
#pragma unroll
            for (int k = 0; k < UNROLL / 2; ++k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(da[k]) : "v"(dc));
        } else if (KIND == 4) {
#pragma unroll
            for (int k = 0; k < UNROLL / 2; ++k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(da[k]) : "v"(dm));
        } else if (KIND == 5) {
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
        } else if (KIND == 6) {
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[k]));
        } else if (KIND == 7) {
#pragma unroll
            for (int k = 0; k < UNROLL / 2; ++k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(ua[k]) : "v"((unsigned)threadIdx.x), "v"(0xD2511F53u) : "vcc");
        } else if (KIND == 8) {
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) asm volatile("v_cmp_ngt_f32 vcc, 0, %0" :: "v"(a[k]) : "vcc");
        } else if (KIND == 9) {
            // fma with one SGPR operand (the render scan's shape)
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "s"(m), "v"(c));
        } else if (KIND == 10) {
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[k]) : "v"(0xD2511F53u));
        } else if (KIND == 11) {
#pragma unroll
            for (int k = 0; k < UNROLL / 2; ++k) asm volatile("v_rcp_f64 %0, %0" : "+v"(da[k]));
        } else if (KIND == 12) {
            // exec = 0 VALU: does the hardware skip them?
            unsigned long long saved;
            asm volatile("s_mov_b64 %0, exec\n s_mov_b64 exec, 0" : "=s"(saved));
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
            asm volatile("s_mov_b64 exec, %0" :: "s"(saved));
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) s += a[k];
#pragma unroll
    for (int k = 0; k < UNROLL / 2; ++k) s += (float)da[k] + pa[k].x + pa[k].y + (float)ua[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char *name, int insts_per_iter, int cus)
{
    float *out; unsigned long long *cyc;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * cus * 8));
    CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * cus * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int bpc : {1, 2, 4, 8}) {           // blocks of 256 threads per CU = waves per SIMD
        int grid = cus * bpc;
        hipLaunchKernelGGL(bench<KIND>, dim3(grid), dim3(256), 0, 0, out, cyc, 1.0f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(bench<KIND>, dim3(grid), dim3(256), 0, 0, out, cyc, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(grid);
        CHECK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost));
        double avg = 0; for (auto v : h) avg += (double)v; avg /= grid;
        double winst = (double)ITERS * insts_per_iter;        // per wave
        // s_memtime / readcyclecounter ticks at a fixed 100 MHz on gfx9 -> use wall time
        double wave_inst_per_s_per_simd = winst * bpc / (ms * 1e-3);
        printf("%-16s waves/SIMD=%d  %8.3f ms  %.3f G wave-inst/s/SIMD  (=> %.2f cycles/wave-inst at 2.4 GHz)  ticks=%.0f\n",
               name, bpc, ms, wave_inst_per_s_per_simd * 1e-9, 2.4e9 / wave_inst_per_s_per_simd, avg);
    }
    CHECK(hipFree(out)); CHECK(hipFree(cyc));
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    int cus = p.multiProcessorCount;
    run<0>("v_fma_f32", UNROLL, cus);
    run<9>("v_fma_f32(sgpr)", UNROLL, cus);
    run<5>("v_sub_f32", UNROLL, cus);
    run<1>("v_pk_fma_f32", UNROLL / 2, cus);
    run<2>("v_fma_f64", UNROLL / 2, cus);
    run<3>("v_add_f64", UNROLL / 2, cus);
    run<4>("v_mul_f64", UNROLL / 2, cus);
    run<6>("v_sqrt_f32", UNROLL, cus);
    run<7>("v_mad_u64_u32", UNROLL / 2, cus);
    run<10>("v_mul_hi_u32", UNROLL, cus);
    run<8>("v_cmp_f32", UNROLL, cus);
    run<11>("v_rcp_f64", UNROLL / 2, cus);
    run<12>("v_fma_f32 exec=0", UNROLL, cus);
    return 0;
}
