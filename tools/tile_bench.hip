// tile_bench.hip -- the MODE 4 tile loop in isolation: B operands by raw buffer loads, two chained
// v_mfma_f32_16x16x32_bf16 per ray group, sign look with max3.  Variants knock parts out to show
// which resource bounds the loop at 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// VAR bit 0: loads, bit 1: MFMA, bit 2: look, bit 3: chained accumulate
template <int VAR>
__global__ __launch_bounds__(256, 4) void k(const uint4 *tab, int tiles, int passes, float *out, unsigned *flag)
{
    extern __shared__ unsigned char lds[];
    const int lane = threadIdx.x & 63;
    bf16x8 A[4][2];
    for (int G = 0; G < 4; ++G) for (int m = 0; m < 2; ++m) {
        uint4 w = make_uint4(0x3c003c00u + lane + G, 0x3c103c10u + m, 0xbc00bc00u, 0x3c003c00u);
        A[G][m] = __builtin_bit_cast(bf16x8, w);
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(tab), 0, (tiles + 1) * 2048, 0x00020000);
    const int voff = lane * 16;
    const f32x4 zero = {0, 0, 0, 0};
    auto ld = [&](int t, int m) -> bf16x8 {
        if (VAR & 1) return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (2 * t + m) * 1024, 0));
        uint4 w = make_uint4(0xbf80bf80u, 0xbf80bf80u + t, 0xbf80bf80u, 0xbf80bf80u);
        asm volatile("" : "+v"(w.x), "+v"(w.y), "+v"(w.z), "+v"(w.w));
        return __builtin_bit_cast(bf16x8, w);
    };
    float sink = 0.0f;
    unsigned hits = 0;
    for (int p = 0; p < passes; ++p) {
        bf16x8 b0 = ld(0, 0), b1 = ld(0, 1);
        for (int t = 0; t < tiles; ++t) {
            const bf16x8 n0 = ld(t + 1, 0), n1 = ld(t + 1, 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 a0, a1;
                if (VAR & 2) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[2 * h][0], b0, zero, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[2 * h + 1][0], b0, zero, 0, 0, 0);
                    if (VAR & 8) {
                        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[2 * h][1], b1, a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[2 * h + 1][1], b1, a1, 0, 0, 0);
                    } else {
                        f32x4 c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[2 * h][1], b1, zero, 0, 0, 0);
                        f32x4 c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[2 * h + 1][1], b1, zero, 0, 0, 0);
                        a0[0] += c0[0]; a1[0] += c1[0];
                    }
                } else {
                    uint4 w0 = __builtin_bit_cast(uint4, b0), w1 = __builtin_bit_cast(uint4, b1);
                    a0 = f32x4{-__uint_as_float(w0.x & 0x7fffffffu), -1.0f, -2.0f, -__uint_as_float(w1.x & 0x7fffffffu)};
                    a1 = f32x4{-__uint_as_float(w0.y & 0x7fffffffu), -1.0f, -2.0f, -__uint_as_float(w1.y & 0x7fffffffu)};
                    asm volatile("" : "+v"(a0), "+v"(a1));
                }
                if (VAR & 4) {
                    const int i0 = __float_as_int(a0[0]), i1 = __float_as_int(a0[1]), i2 = __float_as_int(a0[2]), i3 = __float_as_int(a0[3]);
                    const int j0 = __float_as_int(a1[0]), j1 = __float_as_int(a1[1]), j2 = __float_as_int(a1[2]), j3 = __float_as_int(a1[3]);
                    const int m = max(max(max(i0, i1), i2), max(max(max(i3, j0), j1), max(max(j2, j3), i0)));
                    if (__builtin_expect(__ballot(m >= 0) != 0ull, 0)) { hits++; atomicOr((unsigned *)lds + lane, 1u); }
                } else {
                    sink += a0[0] + a1[3];
                }
            }
            b0 = n0; b1 = n1;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = sink + hits;
    if (hits == 0xffffffffu) *flag = 1;
}

template <int VAR>
void run(const char *name, int cus, const uint4 *d_tab, int tiles, float *d_out, unsigned *d_flag, double ghz)
{
    const int passes = 2000;
    for (int bpc : {1, 2, 3, 4}) {
        size_t lds = (160 * 1024 / bpc) & ~255; if (lds > 65536) lds = 65536;
        CHECK(hipFuncSetAttribute((const void *)k<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        int grid = cus * bpc;
        hipLaunchKernelGGL(k<VAR>, dim3(grid), dim3(256), lds, 0, d_tab, tiles, 50, d_out, d_flag);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<VAR>, dim3(grid), dim3(256), lds, 0, d_tab, tiles, passes, d_out, d_flag);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        double iters_per_simd = (double)bpc * passes * tiles;      // bpc waves per SIMD (4 waves per block, 4 SIMDs)
        printf("%-44s waves/SIMD=%d %8.3f ms  %7.1f cycles per tile-iteration per SIMD (8 MFMA + 2 looks) @%.2f GHz\n",
               name, bpc, ms, ms * 1e-3 * ghz * 1e9 / iters_per_simd, ghz);
    }
}

int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = 2.3;
    const int tiles = 34;
    std::vector<uint4> tab((tiles + 2) * 128);
    for (size_t i = 0; i < tab.size(); ++i) tab[i] = make_uint4(0xbf80bf80u, 0xbf80bf80u, 0xbf80bf80u, 0xbf80bf80u);  // -1.0 everywhere
    uint4 *d_tab; float *d_out; unsigned *d_flag;
    CHECK(hipMalloc(&d_tab, tab.size() * 16)); CHECK(hipMemcpy(d_tab, tab.data(), tab.size() * 16, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_out, cus * 4 * 256 * 4)); CHECK(hipMalloc(&d_flag, 4));
    run<15>("loads + chained MFMA + look (as shipped)", cus, d_tab, tiles, d_out, d_flag, ghz);
    run<14>("no loads", cus, d_tab, tiles, d_out, d_flag, ghz);
    run<7>("loads + unchained MFMA + look", cus, d_tab, tiles, d_out, d_flag, ghz);
    run<11>("loads + chained MFMA, no look", cus, d_tab, tiles, d_out, d_flag, ghz);
    run<10>("chained MFMA only", cus, d_tab, tiles, d_out, d_flag, ghz);
    run<2>("unchained MFMA only", cus, d_tab, tiles, d_out, d_flag, ghz);
    run<5>("loads + look, no MFMA", cus, d_tab, tiles, d_out, d_flag, ghz);
    return 0;
}
