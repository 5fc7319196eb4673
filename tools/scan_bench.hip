// scan_bench.hip -- isolates the sphere-scan inner loop of the render kernel (same
// filter arithmetic, same scalar batch loads) to measure cycles per wave-test under
// different decision schemes.  Development tool; see DESIGN.md section 6.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../rtiow_amd/csrc/rt_device.hpp"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

using namespace rt;
typedef const float __attribute__((address_space(4))) cfloat;

// VARIANT 0: FMAs only, results folded into a running max (no decision)
//         1: v_cmp per test, lane mask OR-ed into a scalar (no branch)
//         2: v_cmp + wave-level branch per test, push to LDS when taken
//         3: v_max over PAIRS, one cmp+branch per pair, re-test inside
//         4: exec-masked push without branch (predication)
//         5: v_max3 over TRIPLES, one cmp+branch per triple
template <int VARIANT>
__global__ __launch_bounds__(256) void scan_kernel(const float *filt_g, int n, int iters, float jitter,
                                                    float *out, unsigned *out_cnt)
{
    extern __shared__ unsigned char dyn_lds[];          // occupancy limiter + candidate lists
    uint16_t (*cand)[256] = reinterpret_cast<uint16_t (*)[256]>(dyn_lds);
    cfloat *filt = (cfloat *)(uintptr_t)filt_g;
    const int tid = threadIdx.x;
    // a camera-like ray per lane
    D3 o = mk(13.0 + 0.001 * tid, 2.0, 3.0);
    D3 d = mk(-13.0 + 0.01 * (tid & 15), -1.7 + 0.01 * (tid >> 4) + jitter, -3.0 + 0.003 * blockIdx.x);
    float acc = -1e30f;
    unsigned long long macc = 0;
    unsigned total = 0;
    for (int it = 0; it < iters; ++it) {
        d.x += 1e-3;                                   // new ray each iteration
        const RayFilter f = make_filter(o, d);
        const RayFilter f2 = make_filter(o, mk(d.x, d.y + 0.37, d.z - 0.2));
        int cnt = 0;
        const int nb = n / 8;
        for (int b = 0; b < nb; ++b) {
            cfloat *q = filt + 32 * (size_t)((VARIANT == 6) ? (b & 0) : b);
            float rec[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) rec[k] = q[k];
            float D[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float cx = rec[4 * k], cy = rec[4 * k + 1], cz = rec[4 * k + 2], kp = rec[4 * k + 3];
                const float hb = __builtin_fmaf(-cx, f.gx, __builtin_fmaf(-cy, f.gy, __builtin_fmaf(-cz, f.gz, f.h0)));
                const float qq = __builtin_fmaf(cx, f.px, __builtin_fmaf(cy, f.py, __builtin_fmaf(cz, f.pz, f.o2)));
                D[k] = __builtin_fmaf(hb, hb, -qq);
                if (VARIANT == 0) {
                    acc = fmaxf(acc, D[k] - kp);
                } else if (VARIANT == 1 || VARIANT == 6) {
                    macc |= __ballot(D[k] >= kp);
                } else if (VARIANT == 7) {
                    macc |= __ballot(D[k] >= kp);
                    const float hb2 = __builtin_fmaf(-cx, f2.gx, __builtin_fmaf(-cy, f2.gy, __builtin_fmaf(-cz, f2.gz, f2.h0)));
                    const float qq2 = __builtin_fmaf(cx, f2.px, __builtin_fmaf(cy, f2.py, __builtin_fmaf(cz, f2.pz, f2.o2)));
                    macc |= __ballot(__builtin_fmaf(hb2, hb2, -qq2) >= kp);
                } else if (VARIANT == 2) {
                    const bool keep = D[k] >= kp;
                    if (__builtin_expect(__ballot(keep) != 0ull, 0)) {
                        if (keep) { cand[cnt & 15][tid] = (uint16_t)(b * 8 + k); cnt++; }
                    }
                } else if (VARIANT == 4) {
                    const bool keep = D[k] >= kp;
                    if (keep) { cand[cnt & 15][tid] = (uint16_t)(b * 8 + k); cnt++; }
                }
            }
            if (VARIANT == 3) {
#pragma unroll
                for (int k = 0; k < 8; k += 2) {
                    const float m = fmaxf(D[k] - rec[4 * k + 3], D[k + 1] - rec[4 * k + 7]);
                    if (__builtin_expect(__ballot(m >= 0.0f) != 0ull, 0)) {
                        if (D[k] >= rec[4 * k + 3]) { cand[cnt & 15][tid] = (uint16_t)(b * 8 + k); cnt++; }
                        if (D[k + 1] >= rec[4 * k + 7]) { cand[cnt & 15][tid] = (uint16_t)(b * 8 + k + 1); cnt++; }
                    }
                }
            }
            if (VARIANT == 5) {
                float e[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) e[k] = D[k] - rec[4 * k + 3];
                const float m = fmaxf(fmaxf(fmaxf(e[0], e[1]), fmaxf(e[2], e[3])), fmaxf(fmaxf(e[4], e[5]), fmaxf(e[6], e[7])));
                if (__builtin_expect(__ballot(m >= 0.0f) != 0ull, 0)) {
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (e[k] >= 0.0f) { cand[cnt & 15][tid] = (uint16_t)(b * 8 + k); cnt++; }
                }
            }
        }
        total += cnt;
    }
    out[blockIdx.x * 256 + tid] = acc + (float)(macc & 0xff);
    atomicAdd(out_cnt, total);
}

template <int V>
void run(const char *name, const float *d_filt, int n, int cus, float *d_out, unsigned *d_cnt)
{
    const int iters = 1500;
    for (int bpc : {4, 5, 8}) {
        size_t lds = (160 * 1024 / bpc) & ~255;             // limits residency to bpc blocks per CU
        if (lds < 8192) lds = 8192;
        CHECK(hipFuncSetAttribute((const void *)scan_kernel<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CHECK(hipMemset(d_cnt, 0, 4));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        int grid = cus * bpc;
        hipLaunchKernelGGL(scan_kernel<V>, dim3(grid), dim3(256), lds, 0, d_filt, n, 20, 0.0f, d_out, d_cnt);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemset(d_cnt, 0, 4));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(scan_kernel<V>, dim3(grid), dim3(256), lds, 0, d_filt, n, iters, 0.0f, d_out, d_cnt);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned cnt; CHECK(hipMemcpy(&cnt, d_cnt, 4, hipMemcpyDeviceToHost));
        double wave_tests = (double)grid * 4 * iters * (n / 8 * 8);
        double per_simd = wave_tests / (cus * 4.0);
        printf("%-28s waves/SIMD=%d  %7.3f ms  %.2f cycles/wave-test/SIMD @2.4GHz  cand/lane/ray=%.2f\n",
               name, bpc, ms, ms * 1e-3 * 2.4e9 / per_simd, (double)cnt / ((double)grid * 256 * iters));
    }
}

int main(int argc, char **argv)
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int n = 528;
    // scene-like table: ground + lattice of small spheres
    std::vector<float> filt(4 * n);
    for (int i = 0; i < n; ++i) {
        double cx, cy, cz, r;
        if (i == 0) { cx = 0; cy = -1000; cz = 0; r = 1000; }
        else { int a = (i % 23) - 11, b = (i / 23) - 11; cx = a + 0.45; cy = 0.2; cz = b + 0.45; r = 0.2; }
        const double KU = (double)kFilterKU, kappa = KU / (1.0 - KU);
        const double c2 = cx * cx + cy * cy + cz * cz, r2 = r * r;
        filt[4 * i] = (float)cx; filt[4 * i + 1] = (float)cy; filt[4 * i + 2] = (float)cz;
        filt[4 * i + 3] = (float)(c2 * (1.0 - kappa) - r2 * (1.0 + 2.0 * kappa));
    }
    float *d_filt, *d_out; unsigned *d_cnt;
    CHECK(hipMalloc(&d_filt, filt.size() * 4));
    CHECK(hipMemcpy(d_filt, filt.data(), filt.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_out, 256 * 8 * p.multiProcessorCount * 4));
    CHECK(hipMalloc(&d_cnt, 4));
    // warm the clocks up: ~0.3 s of back-to-back work before anything is timed
    for (int w = 0; w < 6; ++w) {
        CHECK(hipFuncSetAttribute((const void *)scan_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 20480));
        hipLaunchKernelGGL(scan_kernel<0>, dim3(p.multiProcessorCount * 8), dim3(256), 20480, 0, d_filt, n, 1500, 0.0f, d_out, d_cnt);
    }
    CHECK(hipDeviceSynchronize());
    run<0>("fma only", d_filt, n, p.multiProcessorCount, d_out, d_cnt);
    run<1>("cmp, no branch", d_filt, n, p.multiProcessorCount, d_out, d_cnt);
    run<6>("cmp, no branch, NO RELOAD", d_filt, n, p.multiProcessorCount, d_out, d_cnt);
    run<7>("cmp, no branch, 2 rays/lane", d_filt, n, p.multiProcessorCount, d_out, d_cnt);
    run<2>("cmp + branch per test", d_filt, n, p.multiProcessorCount, d_out, d_cnt);
    run<4>("cmp + predicated push", d_filt, n, p.multiProcessorCount, d_out, d_cnt);
    run<3>("max + branch per pair", d_filt, n, p.multiProcessorCount, d_out, d_cnt);
    run<5>("max + branch per 8", d_filt, n, p.multiProcessorCount, d_out, d_cnt);
    return 0;
}
