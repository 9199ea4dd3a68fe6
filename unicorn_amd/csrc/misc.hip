// Small HBM-bound glue kernels of the path: casts, PixelShuffle, prior pyramid, box decode, position
// embedding resize, aligned-bilinear upsample-add, instance-embedding sampling, CondInst dynamic mask head.
#include "kernels.h"
#include "mask_interp.h"

// ------------------------------------------------------------------------------------------------
// fp32 rows -> GEMM operand rows in the context's operand format (`b32`: FMT_H2 f16x2 hi/lo groups -- the headline --, FMT_F32 or FMT_BF16);
// the element type of `out` is nominal (the historical bf16 pointer type of the operand buffers), act_store8 writes the real format
__global__ void cast_operand_kernel(const float* x, int ldx, bf16* out, int ldo, int M, int C8, int b32) {
    const long total = (long)M * C8;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        int m = (int)(e / C8), c = (int)(e - (long)m * C8) * 8;
        const float4* xp = reinterpret_cast<const float4*>(x + (size_t)m * ldx + c);
        float4 a = xp[0], b = xp[1];
        float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        act_store8(out, (size_t)m * ldo + c, v, b32);
    }
}
int launch_cast_operand(const float* x, int ldx, bf16* out, int ldo, int M, int C, hipStream_t s, int b32) {
    UNI_REQUIRE(C % 8 == 0 && ldx % 4 == 0 && ldo % 8 == 0, "cast_operand: C=%d ldx=%d ldo=%d", C, ldx, ldo);
    long total = (long)M * (C / 8);
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(cast_operand_kernel, dim3(grid), dim3(256), 0, s, x, ldx, out, ldo, M, C / 8, b32);
    return 0;
}

__global__ void cast_operand_pair_kernel(const float* x0, const float* x1, bf16* out, int hw, int C8, int B, int b32) {
    const long total = (long)2 * B * hw * C8;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long row = e / C8;                       // token over [B][2][hw]
        const int c = (int)(e - row * C8) * 8;
        const long t = row / hw;
        const int pix = (int)(row - t * hw);
        const float* src = ((t & 1) ? x1 : x0) + (((t >> 1) * hw + pix) * (long)C8 * 8) + c;
        const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
        const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        act_store8(out, (size_t)row * C8 * 8 + c, v, b32);
    }
}
int launch_cast_operand_pair(const float* x0, const float* x1, bf16* out, int hw, int C, int B, hipStream_t s, int b32) {
    UNI_REQUIRE(C % 8 == 0 && hw > 0 && B > 0, "cast_operand_pair: C=%d hw=%d B=%d", C, hw, B);
    const long total = (long)2 * B * hw * (C / 8);
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(cast_operand_pair_kernel, dim3(grid), dim3(256), 0, s, x0, x1, out, hw, C / 8, B, b32);
    return 0;
}

// PixelShuffle(2) + cast: fp32 NHWC (h,w,C) -> bf16 NHWC (2h,2w,C/4); in channel c*4+dy*2+dx -> out (2y+dy,2x+dx,c)
__global__ void pixel_shuffle_kernel(const float* x, bf16* out, int h, int w, int C, int b32, int B) {
    const int Co = C >> 2;
    const long total = (long)4 * h * w * Co * B;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        int c = (int)(e % Co);
        long pix = e / Co;
        int ox = (int)(pix % (2 * w));
        long oyg = pix / (2 * w);
        int sb = (int)(oyg / (2 * h)), oy = (int)(oyg - (long)sb * 2 * h);
        int y = oy >> 1, xx = ox >> 1, dy = oy & 1, dx = ox & 1;
        act_store1(out, (size_t)e, x[(((size_t)sb * h + y) * w + xx) * C + c * 4 + dy * 2 + dx], b32);
    }
}
int launch_pixel_shuffle_bf16(const float* x, bf16* out, int h, int w, int C, hipStream_t s, int b32, int B) {
    UNI_REQUIRE(C % 4 == 0, "pixel_shuffle: C=%d", C);
    long total = (long)h * w * C * B;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(pixel_shuffle_kernel, dim3(grid), dim3(256), 0, s, x, out, h, w, C, b32, B);
    return 0;
}

// unicorn_sot.py:103-105: bilinear 1/2 and 1/4 (align_corners=False) of the (K,H8,W8) prior
__global__ void prior_pyramid_kernel(const float* p8, float* p16, float* p32, int K, int H8, int W8) {
    const int H16 = H8 / 2, W16 = W8 / 2, H32 = H8 / 4, W32 = W8 / 4;
    const int n16 = K * H16 * W16, n32 = K * H32 * W32;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n16 + n32; e += gridDim.x * blockDim.x) {
        if (e < n16) {
            int x = e % W16, y = (e / W16) % H16, k = e / (W16 * H16);
            const float* s = p8 + ((size_t)k * H8 + 2 * y) * W8 + 2 * x;
            p16[e] = 0.25f * (s[0] + s[1] + s[W8] + s[W8 + 1]);
        } else {
            int f = e - n16;
            int x = f % W32, y = (f / W32) % H32, k = f / (W32 * H32);
            const float* s = p8 + ((size_t)k * H8 + 4 * y + 1) * W8 + 4 * x + 1;
            p32[f] = 0.25f * (s[0] + s[1] + s[W8] + s[W8 + 1]);
        }
    }
}
int launch_prior_pyramid(const float* p8, float* p16, float* p32, int K, int H8, int W8, hipStream_t s) {
    int n = K * ((H8 / 2) * (W8 / 2) + (H8 / 4) * (W8 / 4));
    if (n == 0) return 0;
    hipLaunchKernelGGL(prior_pyramid_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, p8, p16, p32, K, H8, W8);
    return 0;
}

// unicorn_head.py:467-482: xy = (xy + grid) * stride, wh = exp(wh) * stride; in place on (A, nch) rows
__global__ void decode_kernel(float* out, int A0, int W0, int A1, int W1, int A2, int W2, int nch, int B) {
    const int ag = blockIdx.x * blockDim.x + threadIdx.x;
    const int A = A0 + A1 + A2;
    if (ag >= A * B) return;
    const int a = ag % A;
    int loc, W;
    float st;
    if (a < A0) { loc = a; W = W0; st = 8.f; }
    else if (a < A0 + A1) { loc = a - A0; W = W1; st = 16.f; }
    else { loc = a - A0 - A1; W = W2; st = 32.f; }
    const int gy = loc / W, gx = loc - gy * W;
    float* o = out + (size_t)ag * nch;
    o[0] = (o[0] + gx) * st;
    o[1] = (o[1] + gy) * st;
    o[2] = expf(o[2]) * st;
    o[3] = expf(o[3]) * st;
}
int launch_decode(const float* raw, float* out, int A0, int W0, int A1, int W1, int A2, int W2, int nch, hipStream_t s, int B) {
    UNI_REQUIRE(raw == out, "decode: in-place only");
    hipLaunchKernelGGL(decode_kernel, dim3(cdiv((A0 + A1 + A2) * B, 256)), dim3(256), 0, s, out, A0, W0, A1, W1, A2, W2, nch, B);
    return 0;
}

// condinst/comm.py:5-27 aligned_bilinear(src, factor), accumulated into dst (mask_branch.py:80-93): dst += up(src)
__global__ void add_aligned_bilinear_kernel(const float* src, int h, int w, int C, int f, float* dst) {
    const int H = f * h, W = f * w, C4 = C >> 2;
    const long total = (long)H * W * C4;
    src += (size_t)blockIdx.y * h * w * C;                       // sample of the batch
    dst += (size_t)blockIdx.y * H * W * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        int c = (int)(e % C4) * 4;
        long pix = e / C4;
        int x = (int)(pix % W), y = (int)(pix / W);
        int y0, y1, x0, x1;
        float fy, fx;
        ab_coord(y, f, h, y0, y1, fy);
        ab_coord(x, f, w, x0, x1, fx);
        const float4 a = *reinterpret_cast<const float4*>(src + ((size_t)y0 * w + x0) * C + c);
        const float4 b = *reinterpret_cast<const float4*>(src + ((size_t)y0 * w + x1) * C + c);
        const float4 cc = *reinterpret_cast<const float4*>(src + ((size_t)y1 * w + x0) * C + c);
        const float4 d = *reinterpret_cast<const float4*>(src + ((size_t)y1 * w + x1) * C + c);
        float4* o = reinterpret_cast<float4*>(dst + (size_t)pix * C + c);
        float4 v = *o;
        const float w00 = (1 - fy) * (1 - fx), w01 = (1 - fy) * fx, w10 = fy * (1 - fx), w11 = fy * fx;
        v.x += w00 * a.x + w01 * b.x + w10 * cc.x + w11 * d.x;
        v.y += w00 * a.y + w01 * b.y + w10 * cc.y + w11 * d.y;
        v.z += w00 * a.z + w01 * b.z + w10 * cc.z + w11 * d.z;
        v.w += w00 * a.w + w01 * b.w + w10 * cc.w + w11 * d.w;
        *o = v;
    }
}
int launch_add_aligned_bilinear(const float* src, int h, int w, int C, int factor, float* dst, hipStream_t s, int B) {
    UNI_REQUIRE(C % 4 == 0 && factor >= 1, "aligned_bilinear: C=%d factor=%d", C, factor);
    long total = (long)factor * h * factor * w * (C / 4);
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(add_aligned_bilinear_kernel, dim3(grid, B), dim3(256), 0, s, src, h, w, C, factor, dst);
    return 0;
}

// position_encoding.py:25-36: [col_embed(x) | row_embed(y)] on an sz x sz grid, bilinear (align_corners=False)
// to (h,w); output NHWC fp32 [h*w][2*nf].  (unicorn.py:250's same-size bicubic resample is the identity.)
__device__ __forceinline__ void bl_coord(int o, int in, int outn, int& i0, int& i1, float& fr) {
    float src = ((float)in / (float)outn) * (o + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    i1 = i0 + 1 < in - 1 ? i0 + 1 : in - 1;
    fr = src - i0;
}
__global__ void pos_embed_kernel(const float* row, const float* col, int sz, int nf, float* out, int h, int w) {
    const int C = 2 * nf;
    const int total = h * w * C;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        int c = e % C, pix = e / C;
        int x = pix % w, y = pix / w;
        int i0, i1;
        float fr;
        if (c < nf) {
            bl_coord(x, sz, w, i0, i1, fr);
            out[e] = (1 - fr) * col[i0 * nf + c] + fr * col[i1 * nf + c];
        } else {
            bl_coord(y, sz, h, i0, i1, fr);
            out[e] = (1 - fr) * row[i0 * nf + c - nf] + fr * row[i1 * nf + c - nf];
        }
    }
}
int launch_pos_embed(const float* row, const float* col, int sz, int nf, float* out, int h, int w, hipStream_t s) {
    hipLaunchKernelGGL(pos_embed_kernel, dim3(cdiv(h * w * 2 * nf, 256)), dim3(256), 0, s, row, col, sz, nf, out, h, w);
    return 0;
}

// evaluators/mot_evaluator.py:1024-1034 (same lines at :822-827): the box centre in stride-8 pixels, cx = c/s - 0.5, is clamped
// to [0, W-1] and normalised by (W-1) (an align_corners=True style grid), then handed to grid_sample(align_corners=False,
// border): the sampled position is therefore x = clamp(c/s - 0.5, 0, W-1) * W/(W-1) - 0.5, clipped to [0, W-1].  The float
// expression order of the reference lines and of grid_sample's unnormalise is kept.
__global__ void sample_embed_kernel(const float* emb, int H, int W, int C, const float* boxes, int ldbox, int n,
                                    float stride, float* out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * C) return;
    const int c = e % C, i = e / C;
    const float* b = boxes + (size_t)i * ldbox;
    float cx = (b[0] + b[2]) / 2 / stride - 0.5f, cy = (b[1] + b[3]) / 2 / stride - 0.5f;
    const float gx = (fminf(fmaxf(cx, 0.f), (float)(W - 1)) / (float)(W - 1) - 0.5f) * 2.0f;
    const float gy = (fminf(fmaxf(cy, 0.f), (float)(H - 1)) / (float)(H - 1) - 0.5f) * 2.0f;
    float x = ((gx + 1) * W - 1) / 2, y = ((gy + 1) * H - 1) / 2;
    x = fminf(fmaxf(x, 0.f), (float)(W - 1));
    y = fminf(fmaxf(y, 0.f), (float)(H - 1));
    const int x0 = (int)floorf(x), y0 = (int)floorf(y);
    const float fx = x - x0, fy = y - y0;
    const int x1 = x0 + 1 < W ? x0 + 1 : W - 1, y1 = y0 + 1 < H ? y0 + 1 : H - 1;
    const float w11 = (x0 + 1 < W && y0 + 1 < H) ? fx * fy : 0.f;
    const float w01 = (x0 + 1 < W) ? fx * (1 - fy) : 0.f;
    const float w10 = (y0 + 1 < H) ? (1 - fx) * fy : 0.f;
    out[e] = (1 - fx) * (1 - fy) * emb[((size_t)y0 * W + x0) * C + c] + w01 * emb[((size_t)y0 * W + x1) * C + c] +
             w10 * emb[((size_t)y1 * W + x0) * C + c] + w11 * emb[((size_t)y1 * W + x1) * C + c];
}
int launch_sample_embed(const float* emb, int H, int W, int C, const float* boxes, int ldbox, int n, float stride,
                        float* out, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(sample_embed_kernel, dim3(cdiv(n * C, 256)), dim3(256), 0, s, emb, H, W, C, boxes, ldbox, n,
                       stride, out);
    return 0;
}

// unicorn_sot.py:52-53,128-139: rounded-box binary mask at full res -> bilinear 1/8 = mean of the central 2x2
// of each 8x8 block
__global__ void label_map_kernel(const float* box, float* out, int H, int W) {
    const int H8 = H / 8, W8 = W / 8;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= H8 * W8) return;
    int x1 = (int)rintf(box[0]), y1 = (int)rintf(box[1]), x2 = (int)rintf(box[2]), y2 = (int)rintf(box[3]);
    x1 = max(0, min(x1, W)); x2 = max(0, min(x2, W));
    y1 = max(0, min(y1, H)); y2 = max(0, min(y2, H));
    const int x = e % W8, y = e / W8;
    float acc = 0.f;
#pragma unroll
    for (int dy = 3; dy <= 4; ++dy)
#pragma unroll
        for (int dx = 3; dx <= 4; ++dx) {
            int py = 8 * y + dy, px = 8 * x + dx;
            acc += (py >= y1 && py < y2 && px >= x1 && px < x2) ? 0.25f : 0.f;
        }
    out[e] = acc;
}
int launch_label_map_s8(const float* box_xyxy, float* out, int H, int W, hipStream_t s) {
    hipLaunchKernelGGL(label_map_kernel, dim3(cdiv((H / 8) * (W / 8), 256)), dim3(256), 0, s, box_xyxy, out, H, W);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// K6: CondInst dynamic mask head (condinst/dynamic_mask_head.py:172-225, :61-87, :138-156, :159-170;
// utils/boxes.py:138-146).  Three passes, all HBM/LDS-light:
//   A: per-instance 10->8->8->1 MLP over the (H,W) mask-feature map (params in LDS)     -> logits
//   B: RAFT-style convex upsample x r (softmax over 9 taps, 3x3 zero-padded unfold) + sigmoid -> scores
//   C: aligned_bilinear x d_rate of the scores                                          -> (n, d*r*H, d*r*W)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void condinst_mlp_kernel(CondInstArgs p) {
    __shared__ float prm[169];
    const int inst = blockIdx.y;
    for (int i = threadIdx.x; i < 169; i += blockDim.x) prm[i] = p.params[(size_t)inst * p.ldp + i];
    __syncthreads();
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= p.H * p.W) return;
    const float soi_tab[5] = {64.f, 128.f, 256.f, 512.f, 1024.f};
    const float soi = soi_tab[p.inst_lvl[inst]];
    const int y = pix / p.W, x = pix - y * p.W;
    float in[10];
    in[0] = (p.inst_loc[inst * 2] - (x * 8 + 4)) / soi;       // comm.py:30-43 locations = arange*8 + 4
    in[1] = (p.inst_loc[inst * 2 + 1] - (y * 8 + 4)) / soi;
    const float4* mf = reinterpret_cast<const float4*>(p.mask_feats + (size_t)pix * 8);
    float4 a = mf[0], b = mf[1];
    in[2] = a.x; in[3] = a.y; in[4] = a.z; in[5] = a.w; in[6] = b.x; in[7] = b.y; in[8] = b.z; in[9] = b.w;
    const float *w0 = prm, *w1 = prm + 80, *w2 = prm + 144, *b0 = prm + 152, *b1 = prm + 160, *b2 = prm + 168;
    float h0[8], h1[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        float s = b0[o];
#pragma unroll
        for (int i = 0; i < 10; ++i) s += w0[o * 10 + i] * in[i];
        h0[o] = fmaxf(s, 0.f);
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        float s = b1[o];
#pragma unroll
        for (int i = 0; i < 8; ++i) s += w1[o * 8 + i] * h0[i];
        h1[o] = fmaxf(s, 0.f);
    }
    float s = b2[0];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += w2[i] * h1[i];
    p.logits_ws[(size_t)inst * p.H * p.W + pix] = s;
}

constexpr int CU_IC = 8;      // instances per thread: the softmax over the 9 taps of up_masks does not depend on the instance
__global__ __launch_bounds__(256) void condinst_upsample_kernel(CondInstArgs p) {
    // thread = (coarse pixel, sub-position i*r+j) x a chunk of CU_IC instances (blockIdx.y): the tap weights (9 exps) are computed once per
    // chunk instead of once per instance (64 candidates of a MOTS frame: 117 -> ~50 us); per instance the arithmetic and its order are unchanged
    const int rr = p.r * p.r;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.H * p.W * rr) return;
    const int sub = e % rr, pix = e / rr;
    const int y = pix / p.W, x = pix - y * p.W;
    const float* um = p.up_masks + (size_t)pix * 9 * rr + sub;
    float lg[9], mx = -3.0e38f;
#pragma unroll
    for (int t = 0; t < 9; ++t) { lg[t] = um[t * rr]; mx = fmaxf(mx, lg[t]); }
    float wt[9], sum = 0.f;
    int off[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        wt[t] = __expf(lg[t] - mx);
        sum += wt[t];
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        off[t] = (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) ? yy * p.W + xx : -1;
    }
    const int i = sub / p.r, j = sub - i * p.r;
    const int RW = p.r * p.W;
    const int i0 = blockIdx.y * CU_IC, i1 = min(p.n, i0 + CU_IC);
    for (int inst = i0; inst < i1; ++inst) {
        const float* L = p.logits_ws + (size_t)inst * p.H * p.W;
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc += wt[t] * (off[t] >= 0 ? L[off[t]] : 0.f);
        const float logit = acc / sum;
        p.coarse_ws[((size_t)inst * p.r * p.H + (p.r * y + i)) * RW + p.r * x + j] = 1.f / (1.f + __expf(-logit));
    }
}

__global__ __launch_bounds__(256) void condinst_final_kernel(CondInstArgs p) {
    const int h = p.r * p.H, w = p.r * p.W, f = p.d_rate;
    const int Ho = f * h, Wo = f * w;
    const int inst = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= Ho * Wo) return;
    const int y = e / Wo, x = e - y * Wo;
    p.out[(size_t)inst * Ho * Wo + e] = ab_sample(p.coarse_ws + (size_t)inst * h * w, h, w, f, y, x);      // mask_interp.h
}

int launch_condinst(const CondInstArgs& a, hipStream_t s) {
    if (a.n == 0) return 0;
    UNI_REQUIRE(a.r >= 1 && a.d_rate >= 1, "condinst: r=%d d_rate=%d", a.r, a.d_rate);
    const int hw = a.H * a.W, rr = a.r * a.r;
    hipLaunchKernelGGL(condinst_mlp_kernel, dim3(cdiv(hw, 256), a.n), dim3(256), 0, s, a);
    hipLaunchKernelGGL(condinst_upsample_kernel, dim3(cdiv(hw * rr, 256), cdiv(a.n, CU_IC)), dim3(256), 0, s, a);
    if (!a.out) return 0;          // the caller continues from coarse_ws (uni_condinst_masks_u8: fused upsample + resize, mask_post.hip)
    if (a.d_rate == 1) {
        UNI_CHECK_HIP(hipMemcpyAsync(a.out, a.coarse_ws, (size_t)a.n * hw * rr * sizeof(float), hipMemcpyDeviceToDevice, s));
    } else {
        const int ho = a.d_rate * a.r * a.H, wo = a.d_rate * a.r * a.W;
        hipLaunchKernelGGL(condinst_final_kernel, dim3(cdiv(ho * wo, 256), a.n), dim3(256), 0, s, a);
    }
    return 0;
}

// deformable_transformer.py:74,124: query = src + pos + level_embed[lvl] (bf16 operand of the offset/weight Linears)
__global__ void add_pos_kernel(const float* src, const float* pos0, const float* pos1, const float* lvl, bf16* out,
                               int hw, int C, int b32, int B) {
    const int C4 = C >> 2;
    const long total = (long)2 * hw * C4 * B;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        int c = (int)(e % C4) * 4;
        int m = (int)(e / C4);                                   // token over [B][2 frames][hw]
        int l = (m / hw) & 1;
        const float* pos = l ? pos1 : pos0;                      // position embedding is shared by the batch
        float4 a = *reinterpret_cast<const float4*>(src + (size_t)m * C + c);
        float4 b = *reinterpret_cast<const float4*>(pos + (size_t)(m % hw) * C + c);
        float4 d = *reinterpret_cast<const float4*>(lvl + l * C + c);
        act_store4(out, (size_t)m * C + c, a.x + (b.x + d.x), a.y + (b.y + d.y), a.z + (b.z + d.z), a.w + (b.w + d.w), b32);
    }
}
int launch_add_pos_bf16(const float* src, const float* pos0, const float* pos1, const float* lvl, bf16* out, int hw,
                        int C, hipStream_t s, int b32, int B) {
    long total = (long)2 * hw * (C / 4) * B;
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(add_pos_kernel, dim3(grid), dim3(256), 0, s, src, pos0, pos1, lvl, out, hw, C, b32, B);
    return 0;
}
