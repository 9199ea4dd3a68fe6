// Shared device/host helpers for the unicorn_amd HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

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
// GELU: erfc by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7 on erf, ~4x fewer VALU ops than erff); sigmoid / SiLU with the
// hardware reciprocal (1 ulp) instead of the IEEE division sequence.  The exact-fp32 precision mode keeps act_apply.
template <int ACT>
__device__ __forceinline__ float act_fast(float x) {
    if (ACT == ACT_GELU) {
        // x Phi(x) = max(x, 0) - |x| h,  h = erfc(|x| / sqrt2) / 2 = t P(t) exp(-x^2 / 2) / 2,  t = 1 / (1 + p |x| / sqrt2):
        // 11 plain VALU + rcp + exp2 per element (the 1/2 is folded into P, log2 e / 2 into the exp2 argument)
        const float ax = fabsf(x);
        const float u = ax * 0.84932180028801904f;                                  // sqrt(log2(e) / 2)
        const float e = __builtin_amdgcn_exp2f(-u * u);
        const float t = __builtin_amdgcn_rcpf(fmaf(ax, 0.23164188588f, 1.f));       // 0.3275911 / sqrt2
        float q = fmaf(0.5307027145f, t, -0.7265760135f);
        q = fmaf(q, t, 0.7107068705f);
        q = fmaf(q, t, -0.142248368f);
        q = fmaf(q, t, 0.127414796f);
        return fmaf(-ax, q * t * e, fmaxf(x, 0.f));
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

// "act" buffers hold GEMM operands.  Three formats (wave-uniform `fmt`, historically named b32 in the arg structs):
//   FMT_BF16  bf16 elements (2 B)                                   - 1 MFMA per product
//   FMT_F32   fp32 elements (4 B), v_mfma_f32_32x32x2_f32           - exact, 1/16 of the bf16 MFMA rate
//   FMT_H2    "f16x2": every fp32 value x is split into hi = f16(x), lo = f16(x - hi) (22 significand bits); a group of
//             8 consecutive channels occupies 32 B: [8 x hi f16][8 x lo f16], so a 16-B LDS chunk is directly one
//             v_mfma_f32_32x32x16_f16 operand and the GEMM issues hi.hi + hi.lo + lo.hi (3 MFMAs per product, fp32
//             accumulate; the dropped lo.lo term is <= 2^-22 relative).  4 B per element like fp32, so element-index
//             arithmetic (ld, column offsets: multiples of 8) is the fp32 one.  oracle/error_budget.py shows this is the
//             narrowest operand format that keeps box IoU >= 0.999 on every stage (profiles/r02_precision_budget.json).
// idx is in ELEMENTS.
enum ActFmt { FMT_BF16 = 0, FMT_F32 = 1, FMT_H2 = 2 };
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;

// Operand range: |x| <= 65504 is represented to 22 bits; larger magnitudes SATURATE at +-65504 (hi = +-65504, lo = 0) -- never inf / NaN
// (lo is taken from the clamped value: taking it from x would make it overflow to inf beyond 2 x 65504 and poison the hi.lo products).
// UNI_CHECK_SAT / uni_ctx_set_check count saturated operands per stage call (engine.hip: sat_scan_kernel).
__device__ __forceinline__ void h2_split(float x, f16& hi, f16& lo) {
    const float c = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
    hi = (f16)c;
    lo = (f16)(c - (float)hi);
}
__device__ __forceinline__ char* h2_addr(void* base, size_t idx) {      // byte address of the hi half of element idx
    return reinterpret_cast<char*>(base) + (idx >> 3) * 32 + (idx & 7) * 2;
}
__device__ __forceinline__ void act_store4(void* base, size_t idx, float a, float b, float c, float d, int fmt) {
    if (fmt == FMT_F32) {
        f32x4 o = {a, b, c, d};
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + idx) = o;
    } else if (fmt == FMT_H2) {
        const float x4[4] = {a, b, c, d};
        f16x4 h, l;
#pragma unroll
        for (int j = 0; j < 4; ++j) { f16 hh, ll; h2_split(x4[j], hh, ll); h[j] = hh; l[j] = ll; }
        char* p = h2_addr(base, idx);
        *reinterpret_cast<f16x4*>(p) = h;
        *reinterpret_cast<f16x4*>(p + 16) = l;
    } else {
        bf16x4 o = {(bf16)a, (bf16)b, (bf16)c, (bf16)d};
        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(base) + idx) = o;
    }
}
__device__ __forceinline__ void act_store8(void* base, size_t idx, const float (&v)[8], int fmt) {
    if (fmt == FMT_F32) {
        f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
        float* p = reinterpret_cast<float*>(base) + idx;
        *reinterpret_cast<f32x4*>(p) = o0;
        *reinterpret_cast<f32x4*>(p + 4) = o1;
    } else if (fmt == FMT_H2) {
        f16x8 h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) { f16 hh, ll; h2_split(v[j], hh, ll); h[j] = hh; l[j] = ll; }
        char* p = reinterpret_cast<char*>(base) + idx * 4;              // idx % 8 == 0
        *reinterpret_cast<f16x8*>(p) = h;
        *reinterpret_cast<f16x8*>(p + 16) = l;
    } else {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)v[j];
        *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(base) + idx) = o;
    }
}
__device__ __forceinline__ void act_store1(void* base, size_t idx, float a, int fmt) {
    if (fmt == FMT_F32) reinterpret_cast<float*>(base)[idx] = a;
    else if (fmt == FMT_H2) {
        f16 h, l;
        h2_split(a, h, l);
        char* p = h2_addr(base, idx);
        *reinterpret_cast<f16*>(p) = h;
        *reinterpret_cast<f16*>(p + 16) = l;
    } else reinterpret_cast<bf16*>(base)[idx] = (bf16)a;
}
static inline int act_elem_bytes(int fmt) { return fmt == FMT_BF16 ? 2 : 4; }

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Guard for per-DEVICE one-time setup on the launch path (hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device property: a
// `static bool` would leave the second GPU of a process without the opt-in, and an unguarded call is a driver call per launch).
// run(setup) calls `setup` (-> hipError_t) until it has SUCCEEDED once per (guard, current device): the device's bit is set only after the
// success, so a failing opt-in is retried and reported on every call (never a launch with more LDS than the kernel may use), and a second
// thread that arrives while the first is still inside the driver call repeats the idempotent call itself instead of launching behind a
// half-done opt-in.  Devices >= 64 are not tracked: the setup runs on every call there (harmless).
struct DevOnce {
    std::atomic<uint64_t> mask{0};
    template <class F>
    hipError_t run(F&& setup) {
        int d = 0;
        (void)hipGetDevice(&d);
        const bool tracked = d >= 0 && d < 64;
        const uint64_t b = tracked ? 1ull << d : 0ull;
        if (tracked && (mask.load(std::memory_order_acquire) & b)) return hipSuccess;
        const hipError_t e = setup();
        if (e == hipSuccess && tracked) mask.fetch_or(b, std::memory_order_release);
        return e;
    }
};
// the > 64 KiB dynamic-LDS opt-in of one kernel instantiation
static inline hipError_t uni_lds_optin(const void* fn, int bytes) { return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }

// error plumbing (api.cpp owns the storage)
void uni_set_error(const char* fmt, ...);
// opt every listed kernel instantiation into `bytes` of dynamic LDS, once per device (DevOnce::run); `return -1` with an error text on failure
#define UNI_LDS_OPTIN(once, what, bytes, ...)                                                                                   \
    do {                                                                                                                        \
        const hipError_t _oe = (once).run([&]() -> hipError_t {                                                                 \
            const void* _fns[] = {__VA_ARGS__};                                                                                 \
            for (const void* _f : _fns) {                                                                                       \
                const hipError_t _e1 = uni_lds_optin(_f, (int)(bytes));                                                         \
                if (_e1 != hipSuccess) return _e1;                                                                              \
            }                                                                                                                   \
            return hipSuccess;                                                                                                  \
        });                                                                                                                     \
        if (_oe != hipSuccess) {                                                                                                \
            uni_set_error("%s: cannot reserve %d bytes of LDS (%s)", what, (int)(bytes), hipGetErrorString(_oe));               \
            return -1;                                                                                                          \
        }                                                                                                                       \
    } while (0)
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
