// Shared device/host helpers for the Lumina OCR HIP engine (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 bits; all activations are NHWC bf16

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_HSWISH = 2, ACT_HSIGMOID = 3, ACT_SIGMOID = 4, ACT_GELU = 5 };

__device__ __forceinline__ float bf16_to_f32(bf16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

// round-to-nearest-even; inputs here are finite (activations), NaN handling not needed on this path
// fp32 -> bf16, round to nearest even: gfx950's v_cvt_pk_bf16_f32 (one instruction per pair; the integer sequence
// (u + 0x7fff + lsb) >> 16 it replaces gives the same bits for every non-NaN input)
typedef __bf16 bf16x2_hw_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    const __bf16 v = (__bf16)f;
    return __builtin_bit_cast(bf16_t, v);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    bf16x2_hw_t v;
    v[0] = (__bf16)lo; v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}

// erf to 1.5e-7 absolute (Abramowitz & Stegun 7.1.26): a reciprocal, a 5-term Horner polynomial and one v_exp_f32 — 14 instructions
// against ~35 of libm's erff.  The GELU it feeds is rounded to a 16-bit tensor right after (8 or 11 mantissa bits): the definition's
// exact-erf GELU (oracle/nets.py) and this one agree except on rounding ties.
__device__ __forceinline__ float fast_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * ax);
    const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const float e = 1.f - poly * __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    return copysignf(e, x);
}
// sigmoid as v_exp_f32 + v_rcp_f32 (each ~1 ulp) instead of libm expf + an IEEE division (~30 instructions per value); its only user,
// the probability map, is rounded to bf16
__device__ __forceinline__ float fast_sigmoidf(float v) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v)); }
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.f + fast_erf(v * 0.70710678118654752f)); }

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case ACT_RELU: return fmaxf(v, 0.f);
        case ACT_HSWISH: return v * fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.f / 6.f);  // x * relu6(x + 3) * fp32(1/6), no division
        case ACT_HSIGMOID: return fminf(fmaxf(0.2f * v + 0.5f, 0.f), 1.f);
        case ACT_SIGMOID: return fast_sigmoidf(v);
        case ACT_GELU: return gelu_erf(v);  // erf GELU (fast_erf above)
        default: return v;
    }
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device setting: applied once per (current device, kernel), so that a
// second handle on another GPU of the same process gets it too (engine.hip)
hipError_t locr_dyn_lds(const void* kernel, int bytes);

// Bijective XCD-aware remap of a 1-D block id: blocks b and b+8 share an XCD (round-robin
// dispatch), so give each XCD label a contiguous range of logical ids (L2 reuse of halos/weights).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

#define LOCR_CHECK(expr)                                                        \
    do {                                                                        \
        hipError_t _e = (expr);                                                 \
        if (_e != hipSuccess) return locr_fail(eng, #expr, hipGetErrorString(_e)); \
    } while (0)
