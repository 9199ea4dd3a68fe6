// Shared device arithmetic of the mask path, so that the FUSED CondInst -> resized bytes kernel (mask_post.hip: condinst_resize_kernel)
// reproduces the two-pass path (misc.hip: condinst_final_kernel -> mask_post.hip: mask_resize_kernel) bit for bit: both passes are the
// SAME inline functions, each with its own floating-point contraction mode pinned inside the function body (the mode of the including
// file does not matter).
#pragma once
#include "common.h"

// aligned_bilinear (condinst/comm.py:5-27): output index o of a factor-f upsample of n source samples -> the two source samples and the
// fraction (replicate pad + align_corners=True + crop, restated as index arithmetic)
__device__ __forceinline__ void ab_coord(int o, int f, int n, int& i0, int& i1, float& fr) {
    int t = o - f / 2;
    t = t < 0 ? 0 : t;
    // t / f in fp32; for a power-of-two factor (d_rate = 1, 2, 4: every configuration of the reference) the product with the exact
    // reciprocal is the same number, and the reciprocal is loop-invariant (one division per thread instead of one per sample)
    const float pos = (f & (f - 1)) == 0 ? (float)t * (1.0f / (float)f) : (float)t / (float)f;
    int a = (int)pos;
    fr = pos - a;
    i0 = a < n - 1 ? a : n - 1;
    i1 = a + 1 < n - 1 ? a + 1 : n - 1;
}
// value of the factor-f aligned-bilinear upsample of s (rows of w samples) from its coordinates; contraction allowed, exactly as
// condinst_final_kernel has always been compiled (hipcc's default).  ONE definition for the two-pass and the fused kernel.
__device__ __forceinline__ float ab_value(const float* __restrict__ row0, const float* __restrict__ row1, int x0, int x1, float fy, float fx) {
#pragma clang fp contract(fast)
    return (1 - fy) * ((1 - fx) * row0[x0] + fx * row0[x1]) + fy * ((1 - fx) * row1[x0] + fx * row1[x1]);
}
__device__ __forceinline__ float ab_sample(const float* __restrict__ s, int h, int w, int f, int y, int x) {
    int y0, y1, x0, x1;
    float fy, fx;
    ab_coord(y, f, h, y0, y1, fy);
    ab_coord(x, f, w, x0, x1, fx);
    return ab_value(s + y0 * w, s + y1 * w, x0, x1, fy, fx);
}

// ATen UpSample.h: area_pixel_compute_source_index (align_corners = false) + guard_index_and_lambda, fp32; NO contraction: every
// product and sum rounds on its own, like the numpy restatement (oracle/mask_oracle.py)
struct SrcIdx { int i0, i1; float w0, w1; };
__device__ __forceinline__ SrcIdx src_index(int d, int n_in, float rscale) {
#pragma clang fp contract(off)
    float real = rscale * ((float)d + 0.5f) - 0.5f;
    real = fmaxf(real, 0.f);
    int i0 = (int)floorf(real);
    i0 = i0 < n_in - 1 ? i0 : n_in - 1;
    float l1 = fminf(fmaxf(real - (float)i0, 0.f), 1.f);
    SrcIdx s;
    s.i0 = i0;
    s.i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    s.w0 = 1.f - l1;
    s.w1 = l1;
    return s;
}
__device__ __forceinline__ float bilerp4(float a, float b, float c, float d, const SrcIdx& sy, const SrcIdx& sx) {
#pragma clang fp contract(off)
    const float top = sx.w0 * a + sx.w1 * b;
    const float bot = sx.w0 * c + sx.w1 * d;
    return sy.w0 * top + sy.w1 * bot;
}
