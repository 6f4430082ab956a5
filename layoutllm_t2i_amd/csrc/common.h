// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of the GLIGEN denoising path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(2))) _Float16 half2_t;
typedef __attribute__((ext_vector_type(4))) _Float16 half4_t;
typedef __attribute__((ext_vector_type(8))) _Float16 half8_t;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define GL_WAVE 64

// v_mfma_f32_32x32x16_f16: D[i][j] += sum_k A[i][k] * B[j][k]   (both operands "row x k")
//   operand fragment: lane l holds row (l & 31), k = 8*(l >> 5) + [0..7]   (8 halfs = 16 B)
//   accumulator:      lane l, reg r holds D[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l & 31]
__device__ __forceinline__ f32x16 mfma32(half8_t a, half8_t b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// v_rcp_f32 (1 ulp) instead of an IEEE division (v_div_scale / v_rcp / 4 FMA / v_div_fmas / v_div_fixup): the result is
// rounded to fp16 right after, 2^13 times coarser
__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float silu_f(float x) { return x * rcp_fast(1.0f + __expf(-x)); }
// erf via Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. fp32-roundoff class and ~3000x below the fp16
// resolution of the value it feeds): 1 rcp + 1 exp + 7 FMA instead of libm erff's ~50 instructions.  The
// GEGLU epilogue evaluates 8192 of these per 128x128 tile, which at K = 320 cost as much as the MFMA loop.
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = rcp_fast(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float r = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(r, x);
}
// erf-GELU (torch F.gelu default; reference attention.py:45)
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f)); }

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

// single v_max3_f32 (plain fmaxf chains get a canonicalising v_max per MFMA-produced operand)
__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// The value a [hi | lo] split is taken of must be ONE value: hipcc may otherwise clone the expression that produces it (e.g. into the
// branch that stores the lo half) and contract the clones differently; fp16(v1) and v2 - fp16(v2) then disagree about the rounding of the
// hi half whenever v sits on a rounding tie (measured: 30 of 655 360 GEGLU outputs off by one fp16 ulp).  An empty asm pins the value.
__device__ __forceinline__ float pin_value(float v) {
    asm volatile("" : "+v"(v));
    return v;
}

__device__ __forceinline__ uint4 ld16(const void* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void st16(void* p, uint4 v) { *reinterpret_cast<uint4*>(p) = v; }

static inline int gl_cdiv(int a, int b) { return (a + b - 1) / b; }

#define GL_CHECK_LAUNCH()                                   \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return (int)e__;             \
    } while (0)
