// Shared device/host helpers for libvsx (gfx950 only: 64-wide wavefronts, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "vsx.h"

typedef _Float16 half_t;
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// host-side error plumbing (api.cpp)
int vsx_fail(int code, const char* fmt, ...);
int vsx_check_launch(const char* what);

#define VSX_REQUIRE(cond, code, ...)            \
    do {                                        \
        if (!(cond)) return vsx_fail(code, __VA_ARGS__); \
    } while (0)

static inline bool vsx_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

#ifdef __HIPCC__
__device__ __forceinline__ uint4 ld16(const half_t* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void st16(half_t* p, uint4 v) { *reinterpret_cast<uint4*>(p) = v; }
__device__ __forceinline__ h8 as_h8(uint4 v) { return __builtin_bit_cast(h8, v); }
__device__ __forceinline__ uint4 as_u4(h8 v) { return __builtin_bit_cast(uint4, v); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// v_exp + v_rcp (1 ulp) instead of an IEEE division: the result is rounded to fp16 anyway
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the fp16 output rounding): one v_rcp, one v_exp
// and five FMAs instead of the ~40-instruction libm erff — the GEGLU epilogue evaluates it once per output element
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    float poly = 1.061405429f;
    poly = poly * t - 1.453152027f;
    poly = poly * t + 1.421413741f;
    poly = poly * t - 0.284496736f;
    poly = poly * t + 0.254829592f;
    const float e = 1.0f - poly * t * __expf(-ax * ax);
    return copysignf(e, x);
}
__device__ __forceinline__ float gelu_erf_f(float x) {
    return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f));
}
#endif
