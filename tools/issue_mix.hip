// issue_mix.hip -- does scalar work cost a SIMD vector-issue time?  SIMD cycles per loop trip of a stream of 16 cheap vector
// instructions (v_xor_b32) alone and with scalar instructions, not-taken / taken branches, EXEC save/restore pairs or LDS reads
// interleaved, at 1, 2 and 4 waves per SIMD (the render kernel runs 4).  Answers whether the 773 SALU + 186 branches per
// wave-bounce of the render kernel (profiles/r03_rocprofv3_target.json) take issue slots from its 1 749 vector instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int KIND>
__global__ __launch_bounds__(256, 4) void k(int iters, float *out, unsigned sk)
{
    __shared__ unsigned lds[1024];
    const int lane = threadIdx.x & 63;
    lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 256] = 1; lds[threadIdx.x + 512] = 2; lds[threadIdx.x + 768] = 3;
    __syncthreads();
    unsigned x[8];
    for (int i = 0; i < 8; ++i) x[i] = lane + i;
    unsigned y = 0x9E3779B9u * (lane + 1);
    unsigned s0 = sk, s1 = sk + 1, s2 = sk + 2, s3 = sk + 3;
    unsigned l0 = 0, l1 = 0, l2 = 0, l3 = 0;
    const unsigned addr = threadIdx.x * 4;
#define V(i) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(x[i]) : "v"(y));
#define S(r) asm volatile("s_add_u32 %0, %0, 7" : "+s"(r) : : "scc");
#define BNT asm volatile("s_cmp_eq_u32 %0, 0x12345\n\ts_cbranch_scc1 1f\n1:" : : "s"(s0) : "scc");          /* compare + branch, never taken (s0 never equals) */
#define BT  asm volatile("s_cmp_lg_u32 %0, 0x12345\n\ts_cbranch_scc1 1f\n\ts_nop 0\n1:" : : "s"(s0) : "scc"); /* always taken, skips one s_nop */
#define EX  asm volatile("s_and_saveexec_b64 %0, vcc\n\ts_mov_b64 exec, %0" : "=s"(ex) : : "scc");
#define L(r, o) asm volatile("ds_read_b32 %0, %1 offset:" #o : "=v"(r) : "v"(addr));
#define V4(a) V(a) V(a+1) V(a+2) V(a+3)
    unsigned long long ex = 0;
    asm volatile("s_mov_b64 vcc, exec" : : : "vcc");
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) { V4(0) V4(4) V4(0) V4(4) }
        else if (KIND == 1) { V(0) S(s0) V(1) S(s1) V(2) S(s2) V(3) S(s3) V(4) S(s0) V(5) S(s1) V(6) S(s2) V(7) S(s3)
                              V(0) S(s0) V(1) S(s1) V(2) S(s2) V(3) S(s3) V(4) S(s0) V(5) S(s1) V(6) S(s2) V(7) S(s3) }
        else if (KIND == 2) { V(0) V(1) S(s0) V(2) V(3) S(s1) V(4) V(5) S(s2) V(6) V(7) S(s3) V(0) V(1) S(s0) V(2) V(3) S(s1) V(4) V(5) S(s2) V(6) V(7) S(s3) }
        else if (KIND == 3) { V4(0) BNT V4(4) BNT V4(0) BNT V4(4) BNT }
        else if (KIND == 4) { V4(0) BT V4(4) BT V4(0) BT V4(4) BT }
        else if (KIND == 5) { V4(0) EX V4(4) EX V4(0) EX V4(4) EX }
        else if (KIND == 6) { S(s0) S(s1) S(s2) S(s3) S(s0) S(s1) S(s2) S(s3) S(s0) S(s1) S(s2) S(s3) S(s0) S(s1) S(s2) S(s3) }
        else if (KIND == 7) { V4(0) L(l0, 0) V4(4) L(l1, 1024) V4(0) L(l2, 2048) V4(4) L(l3, 3072) asm volatile("s_waitcnt lgkmcnt(0)"); x[0] ^= l0 ^ l1 ^ l2 ^ l3; }
        else if (KIND == 8) { V4(0) V4(4) V4(0) V4(4) S(s0) S(s1) S(s2) S(s3) S(s0) S(s1) S(s2) S(s3) S(s0) S(s1) S(s2) S(s3) S(s0) S(s1) S(s2) S(s3) }
    }
    unsigned r = s0 ^ s1 ^ s2 ^ s3 ^ (unsigned)ex;
    for (int i = 0; i < 8; ++i) r ^= x[i];
    out[blockIdx.x * 256 + threadIdx.x] = (float)r;
}

template <int KIND>
void run(const char *name, int cus, float *d_out)
{
    const int iters = 20000;
    float cyc[3];
    int w = 0;
    for (int bpc : {1, 2, 4}) {
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<KIND>, dim3(cus * bpc), dim3(256), 0, 0, 100, d_out, 3u);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(cus * bpc), dim3(256), 0, 0, iters, d_out, 3u);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        cyc[w++] = (float)(ms * 1e-3 * 2.3e9 / ((double)bpc * iters));
    }
    printf("%-58s %7.1f %7.1f %7.1f   SIMD cycles (at 2.3 GHz) per trip per wave-share at 1, 2, 4 waves/SIMD\n", name, cyc[0], cyc[1], cyc[2]);
}

int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *d_out; CHECK(hipMalloc(&d_out, cus * 4 * 256 * 4));
    run<0>("16 v_xor", cus, d_out);
    run<1>("16 v_xor + 16 s_add interleaved 1:1", cus, d_out);
    run<2>("16 v_xor + 8 s_add interleaved 2:1", cus, d_out);
    run<8>("16 v_xor then 16 s_add (blocks)", cus, d_out);
    run<6>("16 s_add only", cus, d_out);
    run<3>("16 v_xor + 4 (s_cmp + s_cbranch not taken)", cus, d_out);
    run<4>("16 v_xor + 4 (s_cmp + s_cbranch taken over one s_nop)", cus, d_out);
    run<5>("16 v_xor + 4 (s_and_saveexec + s_mov exec)", cus, d_out);
    run<7>("16 v_xor + 4 ds_read_b32 + one wait", cus, d_out);
    return 0;
}
