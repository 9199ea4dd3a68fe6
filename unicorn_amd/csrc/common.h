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
