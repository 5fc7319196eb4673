// rt_device.hpp -- device-side arithmetic of the render megakernel (gfx950).
//
// This file IS the f32 arithmetic contract of DESIGN.md section 4, written for
// the GPU: binary32 everywhere, correctly rounded + - * / sqrt, subnormals
// kept, and multiply-adds fused ONLY where __builtin_fmaf is spelled out
// (the translation unit is compiled with -ffp-contract=off).  Reference lines
// cited are paths under the upstream repo (src/...).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }

// vec3.rs:95-97 / :87-89 as fused chains (contract C2).
__device__ __forceinline__ float dot(V3 a, V3 b)
{
    return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x));
}
__device__ __forceinline__ float len2(V3 a) { return dot(a, a); }

// Correctly rounded by -fhip-fp32-correctly-rounded-divide-sqrt (hipcc default,
// passed explicitly by the build).
__device__ __forceinline__ float rsqrt_len(V3 a) { return 1.0f / __builtin_sqrtf(len2(a)); }
// vec3.rs:107-109 with Div<f64> = multiply by the reciprocal (:371-375).
__device__ __forceinline__ V3 unit_vector(V3 a) { return a * rsqrt_len(a); }

// vec3.rs:116-118:  v - (2*dot(v,n)) * n
__device__ __forceinline__ V3 reflect(V3 v, V3 n) { return v - n * (2.0f * dot(v, n)); }

__device__ __forceinline__ float min_1(float x) { return (x < 1.0f) ? x : 1.0f; }   // 1.0.min(x), NaN -> 1

// vec3.rs:120-125
__device__ __forceinline__ V3 refract(V3 uv, V3 n, float etai_over_etat)
{
    float cos_theta = min_1(-dot(uv, n));
    V3 perp = (uv + n * cos_theta) * etai_over_etat;
    V3 par = n * (-__builtin_sqrtf(__builtin_fabsf(1.0f - len2(perp))));
    return perp + par;
}

// materials.rs:78-82; powi(2) = x*x, powi(5) = ((x*x)*(x*x))*x
__device__ __forceinline__ float reflectance(float cosine, float ref_idx)
{
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    float x = 1.0f - cosine;
    float x2 = x * x;
    float x5 = (x2 * x2) * x;
    return r0 + (1.0f - r0) * x5;
}

// ---- Philox4x32-10 (Random123), counter = (pixel, sample, event, 0) ---------
struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1)
{
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    constexpr uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0;      // v_mad_u64_u32: hi and lo in one go
        uint64_t p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += W0; k1 += W1;
    }
    U4 o; o.x = c0; o.y = c1; o.z = c2; o.w = c3;
    return o;
}

// word -> uniform in [0,1): (w >> 8) * 2^-24, exact in f32.
__device__ __forceinline__ float u01(uint32_t w) { return (float)(w >> 8) * 5.9604644775390625e-08f; }
// gen_range(-1.0..1.0) / (-1.0..=1.0): 2u - 1, exact for 24-bit u.
__device__ __forceinline__ float u11(uint32_t w) { return 2.0f * u01(w) - 1.0f; }

// Contract C5: truncate one radiance channel to the 2^-32 grid.
__device__ __forceinline__ unsigned long long quantize(float x)
{
    if (!(x >= 0.0f)) return 0ull;               // NaN and negatives
    if (x > 1073741824.0f) x = 1073741824.0f;    // 2^30
    return (unsigned long long)((double)x * 4294967296.0);
}

} // namespace rt
