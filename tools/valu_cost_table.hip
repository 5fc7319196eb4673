// valu_cost_table.hip -- SIMD cycles per vector instruction at 4 waves per SIMD (the render kernel's occupancy), every
// wave running the same stream of 16 independent-enough instructions of ONE kind per loop trip (8 rotating registers).
// The render kernel is bound by vector-instruction issue; this table says what each encoding costs there.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define BODY16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

template <int KIND>
__global__ __launch_bounds__(256, 4) void k(int iters, float *out, unsigned sk, double sd, unsigned long long m64)
{
    const int lane = threadIdx.x & 63;
    float x[8];
    double d[8];
    for (int i = 0; i < 8; ++i) { x[i] = (float)(lane + i); d[i] = (double)(lane + i) * 1.0000001; }
    float y = -(float)lane, z = 0.5f;
    double dy = 1.0000001, dz = 1e-9;
    unsigned long long acc64 = lane;
    for (int it = 0; it < iters; ++it) {
#define S_ADD32(i) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(x[i]) : "v"(y));
#define S_ADD64E(i) asm volatile("v_add_f32_e64 %0, %1, %0" : "+v"(x[i]) : "v"(y));
#define S_MAX32(i) asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(x[i]) : "v"(y));
#define S_MAX64(i) asm volatile("v_max_f32_e64 %0, %1, %0" : "+v"(x[i]) : "v"(y));
#define S_MAXABS(i) asm volatile("v_max_f32_e64 %0, |%0|, |%1|" : "+v"(x[i]) : "v"(y));
#define S_MINI(i) asm volatile("v_min_i32_e32 %0, %1, %0" : "+v"(x[i]) : "v"(y));
#define S_MIN3(i) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
#define S_ANDS(i) asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(x[i]) : "s"(sk));
#define S_ANDL(i) asm volatile("v_and_b32_e32 %0, 0x7fffffff, %0" : "+v"(x[i]));
#define S_XOR(i) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(x[i]) : "v"(y));
#define S_XORS(i) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(x[i]) : "s"(sk));
#define S_BITOP3(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(x[i]) : "v"(y), "v"(z));
#define S_BITOP3S(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(x[i]) : "v"(y), "s"(sk));
#define S_MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "+v"(d[i]) : "v"(y), "v"(z) : "vcc");
#define S_CNDMASK(i) asm volatile("v_cndmask_b32_e32 %0, %1, %0, vcc" : "+v"(x[i]) : "v"(y) : );
#define S_CMP(i) asm volatile("v_cmp_ge_f32_e32 vcc, %1, %0" : : "v"(x[i]), "v"(y) : "vcc");
#define S_ADDF64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dz));
#define S_MULF64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dy));
#define S_FMAF64(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dy), "v"(dz));
#define S_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x[i]) : "v"(y));
#define S_MOV(i) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(x[i]) : "v"(y));
#define S_MULF32(i) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(x[i]) : "v"(z));
#define S_FMAF32(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(z), "v"(y));
#define S_CVT(i) asm volatile("v_cvt_f64_u32_e32 %0, %1" : "=v"(d[i]) : "v"(x[i]));
#define S_ADDU(i) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(x[i]) : "v"(y));
#define S_MAX3ABS(i) asm volatile("v_max3_f32 %0, |%0|, |%1|, |%2|" : "+v"(x[i]) : "v"(y), "v"(z));
#define S_MAD64S(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "+v"(d[i]) : "v"(y), "s"(sk) : "vcc");
#define S_BITOP3S0(i) asm volatile("v_bitop3_b32 %0, %2, %0, %1 bitop3:0x96" : "+v"(x[i]) : "v"(y), "s"(sk));
#define S_MULF64S(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "s"(sd));
#define S_ADDF64S(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "s"(sd));
#define S_MADI64(i) asm volatile("v_mad_i64_i32 %0, vcc, %1, %1, %0" : "+v"(d[i]) : "v"(y) : "vcc");
#define S_OR(i) asm volatile("v_or_b32_e32 %0, %1, %0" : "+v"(x[i]) : "v"(y));
#define S_CMPABS(i) asm volatile("v_cmp_lt_f32_e64 vcc, |%0|, 2.0" : : "v"(x[i]) : "vcc");
#define S_CNDE64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "s"(m64));
        if (KIND == 0) { BODY16(S_ADD32) } else if (KIND == 1) { BODY16(S_ADD64E) } else if (KIND == 2) { BODY16(S_MAX32) }
        else if (KIND == 3) { BODY16(S_MAX64) } else if (KIND == 4) { BODY16(S_MAXABS) } else if (KIND == 5) { BODY16(S_MINI) }
        else if (KIND == 6) { BODY16(S_MIN3) } else if (KIND == 7) { BODY16(S_ANDS) } else if (KIND == 8) { BODY16(S_ANDL) }
        else if (KIND == 9) { BODY16(S_XOR) } else if (KIND == 10) { BODY16(S_XORS) } else if (KIND == 11) { BODY16(S_BITOP3) }
        else if (KIND == 12) { BODY16(S_BITOP3S) } else if (KIND == 13) { BODY16(S_MAD64) } else if (KIND == 14) { BODY16(S_CNDMASK) }
        else if (KIND == 15) { BODY16(S_CMP) } else if (KIND == 16) { BODY16(S_ADDF64) } else if (KIND == 17) { BODY16(S_MULF64) }
        else if (KIND == 18) { BODY16(S_FMAF64) } else if (KIND == 19) { BODY16(S_LSHLADD) } else if (KIND == 20) { BODY16(S_MOV) }
        else if (KIND == 21) { BODY16(S_MULF32) } else if (KIND == 22) { BODY16(S_FMAF32) } else if (KIND == 23) { BODY16(S_CVT) }
        else if (KIND == 24) { BODY16(S_ADDU) } else if (KIND == 25) { BODY16(S_MAX3ABS) }
        else if (KIND == 26) { BODY16(S_MAD64S) } else if (KIND == 27) { BODY16(S_BITOP3S0) } else if (KIND == 28) { BODY16(S_MULF64S) }
        else if (KIND == 29) { BODY16(S_ADDF64S) } else if (KIND == 30) { BODY16(S_MADI64) } else if (KIND == 31) { BODY16(S_OR) }
        else if (KIND == 32) { BODY16(S_CMPABS) } else if (KIND == 33) { BODY16(S_CNDE64) }
    }
    float r = (float)acc64;
    for (int i = 0; i < 8; ++i) r += x[i] + (float)d[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int KIND>
void run(const char *name, int cus, float *d_out)
{
    const int iters = 20000;
    float cyc[3];
    int w = 0;
    for (int bpc : {1, 2, 4}) {
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<KIND>, dim3(cus * bpc), dim3(256), 0, 0, 100, d_out, 0x7fffffffu, 1.0000001, 0x5555555555555555ull);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(cus * bpc), dim3(256), 0, 0, iters, d_out, 0x7fffffffu, 1.0000001, 0x5555555555555555ull);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        cyc[w++] = (float)(ms * 1e-3 * 2.3e9 / ((double)bpc * iters * 16));
    }
    printf("%-34s %6.2f %6.2f %6.2f   SIMD cycles per instruction at 1, 2, 4 waves/SIMD\n", name, cyc[0], cyc[1], cyc[2]);
}

int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *d_out; CHECK(hipMalloc(&d_out, cus * 4 * 256 * 4));
    run<0>("v_add_f32_e32 (VOP2)", cus, d_out);
    run<1>("v_add_f32_e64 (VOP3)", cus, d_out);
    run<2>("v_max_f32_e32", cus, d_out);
    run<3>("v_max_f32_e64", cus, d_out);
    run<4>("v_max_f32_e64 |a|,|b|", cus, d_out);
    run<25>("v_max3_f32 |a|,|b|,|c|", cus, d_out);
    run<5>("v_min_i32_e32", cus, d_out);
    run<6>("v_min3_i32", cus, d_out);
    run<7>("v_and_b32_e32 sgpr", cus, d_out);
    run<8>("v_and_b32_e32 literal", cus, d_out);
    run<9>("v_xor_b32_e32", cus, d_out);
    run<10>("v_xor_b32_e32 sgpr", cus, d_out);
    run<11>("v_bitop3_b32", cus, d_out);
    run<12>("v_bitop3_b32 sgpr", cus, d_out);
    run<13>("v_mad_u64_u32", cus, d_out);
    run<26>("v_mad_u64_u32 sgpr multiplier", cus, d_out);
    run<30>("v_mad_i64_i32", cus, d_out);
    run<27>("v_bitop3_b32 sgpr as src0", cus, d_out);
    run<31>("v_or_b32_e32", cus, d_out);
    run<32>("v_cmp_lt_f32_e64 |x|, 2.0", cus, d_out);
    run<33>("v_cndmask_b32_e64 (sgpr mask)", cus, d_out);
    run<14>("v_cndmask_b32_e32", cus, d_out);
    run<15>("v_cmp_ge_f32_e32", cus, d_out);
    run<24>("v_add_u32_e32", cus, d_out);
    run<19>("v_lshl_add_u32", cus, d_out);
    run<20>("v_mov_b32_e32", cus, d_out);
    run<21>("v_mul_f32_e32", cus, d_out);
    run<22>("v_fma_f32", cus, d_out);
    run<23>("v_cvt_f64_u32_e32", cus, d_out);
    run<16>("v_add_f64", cus, d_out);
    run<17>("v_mul_f64", cus, d_out);
    run<28>("v_mul_f64 sgpr operand", cus, d_out);
    run<29>("v_add_f64 sgpr operand", cus, d_out);
    run<18>("v_fma_f64", cus, d_out);
    return 0;
}
