// Shared device/host helpers for the unicorn_amd HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

enum UniAct { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_SILU = 3, ACT_SIGMOID = 4 };

__device__ __forceinline__ float act_apply(float x, int act) {
    switch (act) {
        case ACT_RELU: return fmaxf(x, 0.f);
        case ACT_GELU: return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));   // nn.GELU() (erf form)
        case ACT_SILU: return x / (1.f + __expf(-x));
        case ACT_SIGMOID: return 1.f / (1.f + __expf(-x));
        default: return x;
    }
}

// Activations for the bf16 path with the activation kind as a COMPILE-TIME constant (branch-free inner loops).
// GELU: erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, ~3x fewer VALU ops than erff); sigmoid / SiLU with the
// hardware reciprocal (1 ulp) instead of the IEEE division sequence.  The exact-fp32 precision mode keeps act_apply.
template <int ACT>
__device__ __forceinline__ float act_fast(float x) {
    if (ACT == ACT_GELU) {
        const float z = fabsf(x) * 0.70710678118654752f;
        const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
        float q = fmaf(1.061405429f, t, -1.453152027f);
        q = fmaf(q, t, 1.421413741f);
        q = fmaf(q, t, -0.284496736f);
        q = fmaf(q, t, 0.254829592f);
        const float e = 1.f - q * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);   // erf(|x|/sqrt2)
        return 0.5f * x * (1.f + copysignf(e, x));
    }
    if (ACT == ACT_RELU) return fmaxf(x, 0.f);
    if (ACT == ACT_SILU) return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
    if (ACT == ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
    return x;
}
__device__ __forceinline__ float act_apply_fast(float x, int act) {
    switch (act) {
        case ACT_GELU: return act_fast<ACT_GELU>(x);
        case ACT_RELU: return act_fast<ACT_RELU>(x);
        case ACT_SILU: return act_fast<ACT_SILU>(x);
        case ACT_SIGMOID: return act_fast<ACT_SIGMOID>(x);
        default: return x;
    }
}

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

// "act" buffers hold GEMM operands: bf16 (es=2) in the default precision, fp32 (es=4) in the exact-fp32 mode.
// idx is in ELEMENTS; b32 is wave-uniform.
__device__ __forceinline__ void act_store4(void* base, size_t idx, float a, float b, float c, float d, int b32) {
    if (b32) {
        f32x4 o = {a, b, c, d};
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + idx) = o;
    } else {
        bf16x4 o = {(bf16)a, (bf16)b, (bf16)c, (bf16)d};
        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(base) + idx) = o;
    }
}
__device__ __forceinline__ void act_store8(void* base, size_t idx, const float (&v)[8], int b32) {
    if (b32) {
        f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
        float* p = reinterpret_cast<float*>(base) + idx;
        *reinterpret_cast<f32x4*>(p) = o0;
        *reinterpret_cast<f32x4*>(p + 4) = o1;
    } else {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)v[j];
        *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(base) + idx) = o;
    }
}
__device__ __forceinline__ void act_store1(void* base, size_t idx, float a, int b32) {
    if (b32) reinterpret_cast<float*>(base)[idx] = a;
    else reinterpret_cast<bf16*>(base)[idx] = (bf16)a;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// error plumbing (api.cpp owns the storage)
void uni_set_error(const char* fmt, ...);
#define UNI_CHECK_HIP(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            uni_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return -2;                                                                        \
        }                                                                                     \
    } while (0)
#define UNI_REQUIRE(cond, ...)              \
    do {                                    \
        if (!(cond)) {                      \
            uni_set_error(__VA_ARGS__);     \
            return -1;                      \
        }                                   \
    } while (0)
