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
// Phi(x) = 0.5 (1 + erf(x / sqrt 2)) by Abramowitz & Stegun 7.1.28, erf z = 1 - (1 + a1 z + ... + a6 z^6)^-16 (|error| <= 3e-7,
// far below the fp16 output rounding; the sqrt 2 is folded into the coefficients): six FMAs, four squarings and ONE
// transcendental.  The GEGLU epilogue evaluates it once per output element and is bound by VALU issue; the first fast form
// (7.1.26: one v_rcp AND one v_exp, each a quarter-rate instruction) spent half of its cycles in the two of them.
__device__ __forceinline__ float norm_cdf_fast(float x) {
    const float ax = fabsf(x);
    float p = 5.3829750000e-06f;
    p = __builtin_fmaf(p, ax, 4.8890635643e-05f);
    p = __builtin_fmaf(p, ax, 3.8003575000e-05f);
    p = __builtin_fmaf(p, ax, 3.2776263241e-03f);
    p = __builtin_fmaf(p, ax, 2.1141006150e-02f);
    p = __builtin_fmaf(p, ax, 4.9867346967e-02f);
    p = __builtin_fmaf(p, ax, 1.0f);
    p *= p;
    p *= p;
    p *= p;
    p *= p;                                                     // overflows to +inf beyond |x| ~ 21: rcp -> 0
    const float hr = 0.5f * __builtin_amdgcn_rcpf(p);           // 0.5 (1 - erf(|x| / sqrt 2)), relative accuracy ~1e-6
    return x < 0.f ? hr : 1.0f - hr;
}
// Two at a time, written on 2-vectors so that the FMAs, squarings and the final combination become packed-fp32 instructions
// (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth per issue slot): gelu(x) = 0.5 x + |x| (0.5 - hr) with hr as above — no
// compare, no select.  Every GEGLU path (persistent and tile kernels, vsx_geglu_fwd) evaluates this form, so they agree bit for bit.
typedef float vsx_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ vsx_f2 gelu_erf_f2(vsx_f2 x) {
    vsx_f2 ax;
    ax[0] = fabsf(x[0]);
    ax[1] = fabsf(x[1]);
    vsx_f2 p = 5.3829750000e-06f;
    p = p * ax + 4.8890635643e-05f;
    p = p * ax + 3.8003575000e-05f;
    p = p * ax + 3.2776263241e-03f;
    p = p * ax + 2.1141006150e-02f;
    p = p * ax + 4.9867346967e-02f;
    p = p * ax + 1.0f;
    p *= p;
    p *= p;
    p *= p;
    p *= p;
    vsx_f2 r;
    r[0] = __builtin_amdgcn_rcpf(p[0]);
    r[1] = __builtin_amdgcn_rcpf(p[1]);
    const vsx_f2 w = r * -0.5f + 0.5f;                          // 0.5 - hr = 0.5 erf(|x| / sqrt 2)
    return ax * w + x * 0.5f;
}
__device__ __forceinline__ float gelu_erf_f(float x) { return gelu_erf_f2(vsx_f2{x, x})[0]; }   // same arithmetic (edge paths)
#endif
