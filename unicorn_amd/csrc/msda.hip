// K3: multi-scale deformable attention forward (replaces the reference's only hand-written CUDA op,
// unicorn/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299, bilinear :33-84).
// Two kernels: `msda_kernel`, the reference-compatible general op behind uni_msda_fwd (any N / heads / levels / points:
// one lane per (query, head, channel), a corner fetch = one contiguous D*4-byte segment), and `msda_wave_kernel`, the
// engine's fused version for Unicorn's fixed geometry (one wave per (token, head), cross-group shuffle reductions).
// value is 8 MB per frame pair at 800x1280 -> L2/MALL resident; the op is bound by L2 gather bandwidth, no LDS needed.
#include "kernels.h"
#include <cstdlib>

struct MsdaShapes { int H[8], W[8], start[8]; };

__device__ __forceinline__ float msda_bilinear(const float* v, int H, int W, int stride, float h, float w) {
    // ms_deform_im2col_cuda.cuh:33-84: corners outside the map contribute 0
    const int h0 = (int)floorf(h), w0 = (int)floorf(w);
    const float lh = h - h0, lw = w - w0, hh = 1.f - lh, hw = 1.f - lw;
    const int h1 = h0 + 1, w1 = w0 + 1;
    float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
    if (h0 >= 0 && w0 >= 0) v1 = v[(size_t)(h0 * W + w0) * stride];
    if (h0 >= 0 && w1 <= W - 1) v2 = v[(size_t)(h0 * W + w1) * stride];
    if (h1 <= H - 1 && w0 >= 0) v3 = v[(size_t)(h1 * W + w0) * stride];
    if (h1 <= H - 1 && w1 <= W - 1) v4 = v[(size_t)(h1 * W + w1) * stride];
    return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;
}

__global__ __launch_bounds__(256) void msda_kernel(const float* __restrict__ value, const float* __restrict__ loc,
                                                   const float* __restrict__ attn, float* __restrict__ out,
                                                   MsdaShapes shp, int N, int S, int M, int D, int Lq, int L, int P) {
    const long total = (long)N * Lq * M * D;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int d = idx % D;
    const int m = (idx / D) % M;
    const int q = (idx / ((long)D * M)) % Lq;
    const int n = idx / ((long)D * M * Lq);
    const float* lp = loc + (((size_t)n * Lq + q) * M + m) * L * P * 2;
    const float* ap = attn + (((size_t)n * Lq + q) * M + m) * L * P;
    const int stride = M * D;
    float acc = 0.f;
    for (int l = 0; l < L; ++l) {
        const int H = shp.H[l], W = shp.W[l];
        const float* vb = value + ((size_t)n * S + shp.start[l]) * stride + m * D + d;
        for (int pt = 0; pt < P; ++pt) {
            const float x = lp[(l * P + pt) * 2] * W - 0.5f;
            const float y = lp[(l * P + pt) * 2 + 1] * H - 0.5f;
            const float wgt = ap[l * P + pt];
            if (y > -1 && x > -1 && y < H && x < W) acc += wgt * msda_bilinear(vb, H, W, stride, y, x);
        }
    }
    out[idx] = acc;
}

int launch_msda(const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc, const float* attn,
                float* out, int N, int S, int M, int D, int Lq, int L, int P, hipStream_t s) {
    UNI_REQUIRE(L >= 1 && L <= 8, "msda: n_levels=%d unsupported (1..8)", L);
    MsdaShapes shp;
    long tot = 0;
    for (int l = 0; l < L; ++l) {
        shp.H[l] = (int)shapes[2 * l];
        shp.W[l] = (int)shapes[2 * l + 1];
        shp.start[l] = (int)lsi[l];
        tot += shapes[2 * l] * shapes[2 * l + 1];
    }
    UNI_REQUIRE(tot == S, "msda: sum(H*W)=%ld != S=%d", tot, S);   // ms_deform_attn.py:94
    const long total = (long)N * Lq * M * D;
    if (total == 0) return 0;
    hipLaunchKernelGGL(msda_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, value, loc, attn, out, shp,
                       N, S, M, D, Lq, L, P);
    return 0;
}

// Engine variant for Unicorn's fixed geometry (8 heads x 32 ch, 2 levels = ref / cur frame of identical (h,w), 4 points): fuses
// ms_deform_attn.py:98-105 (softmax over the 8 logits, loc = ref + off/(W,H)) and deformable_transformer.py:141-153 (reference
// points) into the sampler; emits the operand format of output_proj.
// CDNA4 shape of the same op: ONE WAVE per (token, head).  The 64 lanes are 8 groups x 8 lanes: group g owns sampling point g
// (level g >> 2, point g & 3), lane j of a group owns channels 4j..4j+3 of the 32-channel head row, so a corner fetch of a
// group is one 128-byte row as 8 x float4.  The softmax over the 8 logits and the final sum over the 8 points are butterfly
// reductions ACROSS the groups (xor 8 / 16 / 32 shuffles); location / bilinear math runs once per point (8-fold instead of
// 32-fold redundancy), 4 float4 loads per lane instead of 32 scalar ones.  Semantics = ms_deform_im2col_cuda.cuh:237-299
// (zero padding per corner, sample skipped unless -1 < x < W and -1 < y < H), ms_deform_attn.py:98-105.
__device__ __forceinline__ float xgroup_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 8, 64)); v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float xgroup_sum(float v) {
    v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64);
}
__global__ __launch_bounds__(256) void msda_wave_kernel(MsdaFusedArgs p) {
    const int hw = p.h * p.w, Lq = 2 * hw;
    const int lane = threadIdx.x & 63;
    const long task = (long)blockIdx.x * 4 + (threadIdx.x >> 6);          // (token, head)
    if (task >= (long)Lq * 8 * p.B) return;                                 // wave-uniform
    const int m = (int)(task & 7);
    const int qg = (int)(task >> 3);                                        // token over [B][2 frames][hw]
    const int sb = qg / Lq, q = qg - sb * Lq;
    const int g = lane >> 3, j = lane & 7;
    const float* row = p.offaw + (size_t)qg * p.ldo;
    const float ox = row[m * 16 + g * 2], oy = row[m * 16 + g * 2 + 1];
    const float lg = row[128 + m * 8 + g];
    const float mx = xgroup_max(lg);
    const float e = __expf(lg - mx);
    const float wgt = e / xgroup_sum(e);
    const int pos = q % hw, i0 = pos / p.w, j0 = pos - i0 * p.w;
    const float lx = (j0 + 0.5f) / p.w + ox / p.w, ly = (i0 + 0.5f) / p.h + oy / p.h;
    const float x = lx * p.w - 0.5f, y = ly * p.h - 0.5f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (y > -1 && x > -1 && y < p.h && x < p.w) {
        const int y0 = (int)floorf(y), x0 = (int)floorf(x);
        const float ly1 = y - y0, lx1 = x - x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        const float* vb = p.value + ((size_t)sb * Lq + (size_t)(g >> 2) * hw) * 256 + m * 32 + j * 4;
        const bool yok0 = y0 >= 0, yok1 = y0 + 1 <= p.h - 1, xok0 = x0 >= 0, xok1 = x0 + 1 <= p.w - 1;
        f32x4 v1 = {0.f, 0.f, 0.f, 0.f}, v2 = v1, v3 = v1, v4 = v1;
        if (yok0 && xok0) v1 = *reinterpret_cast<const f32x4*>(vb + (size_t)(y0 * p.w + x0) * 256);
        if (yok0 && xok1) v2 = *reinterpret_cast<const f32x4*>(vb + (size_t)(y0 * p.w + x0 + 1) * 256);
        if (yok1 && xok0) v3 = *reinterpret_cast<const f32x4*>(vb + (size_t)((y0 + 1) * p.w + x0) * 256);
        if (yok1 && xok1) v4 = *reinterpret_cast<const f32x4*>(vb + (size_t)((y0 + 1) * p.w + x0 + 1) * 256);
        // same association as the reference kernel: w1 v1 + w2 v2 + w3 v3 + w4 v4, then times the attention weight
        acc = (ly0 * lx0) * v1 + (ly0 * lx1) * v2 + (ly1 * lx0) * v3 + (ly1 * lx1) * v4;
        acc *= wgt;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = xgroup_sum(acc[c]);
    if (g == 0) act_store4(p.out, (size_t)qg * 256 + m * 32 + j * 4, acc[0], acc[1], acc[2], acc[3], p.b32);
}

int launch_msda_fused(const MsdaFusedArgs& a, hipStream_t s) {
    const long tasks = (long)2 * a.h * a.w * 8 * a.B;
    hipLaunchKernelGGL(msda_wave_kernel, dim3((unsigned)((tasks + 3) / 4)), dim3(256), 0, s, a);
    return 0;
}
