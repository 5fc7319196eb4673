// tube_loop_bench.hip -- the MODE 5 tile loop of rt_kernels.hpp in isolation: per 32-sphere tile one B operand +
// bound by raw buffer loads (a tile ahead), four v_mfma_f32_32x32x16_bf16 (one per 16-ray group, two results in
// flight), the 14-instruction look after each.  Variants knock parts out / reorder the look to show what bounds
// the loop at 1..4 waves per SIMD.  Prints SIMD cycles per (MFMA + look) unit.
//   VAR bit 0: loads   bit 1: MFMA   bit 2: look   bit 3: look written as two interleaved half-trees
//       bit 4: the look reads registers the MFMAs do NOT write (no MFMA -> VALU dependency, no wait states)
//       bit 5: no wave-level branch after the look (the minimum is folded into a running minimum)
//       bit 6: three results in flight instead of two
//       bit 7: the look as bit logic: columns pre-scaled so that "kept" is |h| < 2, i.e. bit 30 of the f32 clear;
//              X = AND over rays of (h1 | h2) in 1 v_or + 7 v_bitop3, one |X| < 2.0 compare
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int VAR>
__global__ __launch_bounds__(256, 4) void k(const uint4 *btab, const float *rtab, int tiles, int passes, float *out, unsigned *flag)
{
    const int lane = threadIdx.x & 63;
    bf16x8 A[4];
    for (int G = 0; G < 4; ++G) {
        uint4 w = make_uint4(0x3c003c00u + lane + G, 0x3c103c10u, 0xbc00bc00u, 0x3c003c00u);
        A[G] = __builtin_bit_cast(bf16x8, w);
    }
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(btab), 0, (tiles + 2) * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(rtab), 0, (tiles + 2) * 128, 0x00020000);
    const int voff = lane * 16, roff = (lane & 31) * 4;
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto load_b = [&](int t) -> bf16x8 {
        if (VAR & 1) return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(brs, voff, t * 1024, 0));
        uint4 w = make_uint4(0x40804080u, 0x40804080u + t, 0x40804080u, 0x40804080u);
        asm volatile("" : "+v"(w.x), "+v"(w.y), "+v"(w.z), "+v"(w.w));
        return __builtin_bit_cast(bf16x8, w);
    };
    auto load_r = [&](int t) -> float {
        if (VAR & 1) return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, roff, t * 128, 0));
        float r = 1e-30f; asm volatile("" : "+v"(r)); return r;
    };
    unsigned hits = 0;
    float run_min = 3e38f;
    f32x16 stale = zero16; stale[0] = 3.0f; stale[5] = -2.0f; stale[9] = 1.5f; asm volatile("" : "+v"(stale));
    auto look = [&](f32x16 acc_in, float bound) {
        if (!(VAR & 4)) { asm volatile("" :: "v"(acc_in)); return; }
        if (VAR & 128) {
            unsigned X = __float_as_uint(acc_in[0]) | __float_as_uint(acc_in[4]);
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (bb + j) X = __builtin_amdgcn_bitop3_b32(X, __float_as_uint(acc_in[8 * bb + j]), __float_as_uint(acc_in[8 * bb + 4 + j]), 0xE0);   // X & (h1 | h2)
            if (__builtin_expect(__ballot(__builtin_fabsf(__uint_as_float(X)) < 2.0f) != 0ull, 0)) hits += 1u;
            return;
        }
        if (VAR & 256) {        // the same as two independent chains of four rays, joined by one AND
            unsigned Xa = __float_as_uint(acc_in[0]) | __float_as_uint(acc_in[4]);
            unsigned Xb = __float_as_uint(acc_in[8]) | __float_as_uint(acc_in[12]);
#pragma unroll
            for (int j = 1; j < 4; ++j) {
                Xa = __builtin_amdgcn_bitop3_b32(Xa, __float_as_uint(acc_in[j]), __float_as_uint(acc_in[4 + j]), 0xE0);
                Xb = __builtin_amdgcn_bitop3_b32(Xb, __float_as_uint(acc_in[8 + j]), __float_as_uint(acc_in[12 + j]), 0xE0);
            }
            if (__builtin_expect(__ballot(__builtin_fabsf(__uint_as_float(Xa & Xb)) < 2.0f) != 0ull, 0)) hits += 1u;
            return;
        }
        f32x16 acc = acc_in;
        if (VAR & 16) { asm volatile("" :: "v"(acc_in)); acc = stale; }
        int m[2][4];
        const int tok = __float_as_int(acc[0]) & 0x7FFFFFFF;
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                asm("v_max_f32_e64 %0, |%1|, |%2|" : "=v"(m[bb][j]) : "v"(acc[8 * bb + j]), "v"(acc[8 * bb + 4 + j]), "v"(tok));
        float nall;
        if (VAR & 8) {      // two independent half-trees, joined at the end (shorter dependent chain)
            const int a = min(min(m[0][0], m[0][1]), m[0][2]), b = min(min(m[1][0], m[1][1]), m[1][2]);
            const int a2 = min(a, m[0][3]), b2 = min(b, m[1][3]);
            nall = __int_as_float(min(a2, b2));
        } else {
            const int n01 = min(min(m[0][0], m[0][1]), m[0][2]);
            const int n02 = min(min(n01, m[0][3]), m[1][0]);
            const int n03 = min(min(n02, m[1][1]), m[1][2]);
            nall = __int_as_float(min(n03, m[1][3]));
        }
        if (VAR & 32) run_min = __builtin_fminf(run_min, nall - bound);
        else if (__builtin_expect(__ballot(nall <= bound) != 0ull, 0)) hits += 1u;
    };
    auto mfma = [&](const bf16x8 &a, const bf16x8 &b) -> f32x16 {
        if (VAR & 2) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, zero16, 0, 0, 0);
        f32x16 r = zero16; r[0] = 3.0f; r[5] = -2.0f; asm volatile("" : "+v"(r)); return r;
    };
    auto do_tile = [&](const bf16x8 &b, float bound) {
        if (VAR & 64) {
            f32x16 a0 = mfma(A[0], b), a1 = mfma(A[1], b), a2 = mfma(A[2], b);
            asm volatile("" : "+v"(a1), "+v"(a2));
            look(a0, bound);
            a0 = mfma(A[3], b);
            asm volatile("" : "+v"(a0));
            look(a1, bound); look(a2, bound); look(a0, bound);
            return;
        }
        f32x16 acc0 = mfma(A[0], b);
        f32x16 acc1 = mfma(A[1], b);
        asm volatile("" : "+v"(acc1));
        look(acc0, bound);
        acc0 = mfma(A[2], b);
        asm volatile("" : "+v"(acc0));
        look(acc1, bound);
        acc1 = mfma(A[3], b);
        asm volatile("" : "+v"(acc1));
        look(acc0, bound);
        look(acc1, bound);
    };
    for (int p = 0; p < passes; ++p) {
        bf16x8 bp = load_b(0), bq;
        float rp = load_r(0), rq;
        int w = 0;
        for (; w + 1 < tiles; w += 2) {
            bq = load_b(w + 1); rq = load_r(w + 1);
            do_tile(bp, rp);
            bp = load_b(w + 2); rp = load_r(w + 2);
            do_tile(bq, rq);
        }
        if (w < tiles) do_tile(bp, rp);
    }
    if (hits == 0xFFFFFFFFu || run_min == 12345.0f) *flag = 1;
    out[blockIdx.x * 256 + threadIdx.x] = (float)hits;
}

template <int VAR>
void run(const char *name, int cus, const uint4 *btab, const float *rtab, int tiles, float *d_out, unsigned *d_flag)
{
    for (int bpc : {1, 2, 4}) {
        const int passes = 600;
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<VAR>, dim3(cus * bpc), dim3(256), 0, 0, btab, rtab, tiles, 4, d_out, d_flag);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<VAR>, dim3(cus * bpc), dim3(256), 0, 0, btab, rtab, tiles, passes, d_out, d_flag);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double units_per_simd = (double)bpc * passes * tiles * 4;      // one wave of each block per SIMD
        printf("%-34s tiles %4d waves/SIMD %d: %8.3f ms  %6.1f cycles per unit @2.3GHz\n", name, tiles, bpc, ms, ms * 1e-3 * 2.3e9 / units_per_simd);
    }
}

int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    for (int tiles : {17}) {
        std::vector<uint4> b((tiles + 2) * 64, make_uint4(0x40804080u, 0x40804080u, 0x40804080u, 0x3f803f80u));
        std::vector<float> r((tiles + 2) * 32, 1e-30f);
        uint4 *d_b; float *d_r, *d_out; unsigned *d_flag;
        CHECK(hipMalloc(&d_b, b.size() * 16)); CHECK(hipMalloc(&d_r, r.size() * 4));
        CHECK(hipMalloc(&d_out, cus * 4 * 256 * 4)); CHECK(hipMalloc(&d_flag, 4));
        CHECK(hipMemcpy(d_b, b.data(), b.size() * 16, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_r, r.data(), r.size() * 4, hipMemcpyHostToDevice));
        run<7>("loads + MFMA + look", cus, d_b, d_r, tiles, d_out, d_flag);
        run<7 + 128>("  bit-logic look", cus, d_b, d_r, tiles, d_out, d_flag);
        run<7 + 256>("  bit-logic look, two chains", cus, d_b, d_r, tiles, d_out, d_flag);
        run<7 + 64>("  three results in flight", cus, d_b, d_r, tiles, d_out, d_flag);
        run<3>("loads + MFMA (no look)", cus, d_b, d_r, tiles, d_out, d_flag);
    }
    return 0;
}
