// HBM-bound normalisation / depthwise kernels of the path (K1a and the LN/GN pieces of K2):
//   layernorm_kernel   : row LN over C (convnext.py:176-184, nn.LayerNorm in deformable_transformer.py:98-119)
//   gn_apply_kernel    : GroupNorm(G) from group sums + SiLU/ReLU (+ prior*beta fusion, unicorn_head.py:272-277)
//   dwconv7_ln_kernel  : depthwise 7x7 + bias + LN fused (convnext.py:43-47)
//   stem_kernel        : conv4x4/s4 + LN_cf on the NCHW input image (convnext.py:76-79)
// All statistics in fp32 with a two-pass (mean, then centred variance) reduction; deterministic
// (no float atomics).  NHWC everywhere so a wave's lanes walk the channel axis (coalesced 16-B accesses).
#include "kernels.h"
#include <cstdlib>

// ------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, the row lives in registers (C <= 1536 -> <= 6 float4 per lane)
// ------------------------------------------------------------------------------------------------
template <int NI>
__global__ __launch_bounds__(256) void layernorm_kernel(LnArgs p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    const int C4 = p.C >> 2;
    const float4* x = reinterpret_cast<const float4*>(p.x + (size_t)row * p.ldx);
    float4 v[NI];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        int idx = lane + 64 * i;
        v[i] = idx < C4 ? x[idx] : make_float4(0, 0, 0, 0);
        s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    const float mean = wave_sum(s) / p.C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        int idx = lane + 64 * i;
        if (idx < C4) {
            float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += a * a + b * b + c * c + d * d;
        }
    }
    const float rstd = 1.f / sqrtf(wave_sum(q) / p.C + p.eps);
    const float4* g4 = reinterpret_cast<const float4*>(p.gamma);
    const float4* b4 = reinterpret_cast<const float4*>(p.beta);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        int idx = lane + 64 * i;
        if (idx < C4) {
            float4 g = g4[idx], b = b4[idx], o;
            o.x = (v[i].x - mean) * rstd * g.x + b.x;
            o.y = (v[i].y - mean) * rstd * g.y + b.y;
            o.z = (v[i].z - mean) * rstd * g.z + b.z;
            o.w = (v[i].w - mean) * rstd * g.w + b.w;
            if (p.outF) {
                if (p.pair_hw) {     // tokens [B][2][hw] -> frame maps [B][hw]
                    const int t = row / p.pair_hw, pix = row - t * p.pair_hw;
                    float* dst = (t & 1) ? p.outF2 : p.outF;
                    *reinterpret_cast<float4*>(dst + ((size_t)(t >> 1) * p.pair_hw + pix) * p.ldf + idx * 4) = o;
                } else {
                    *reinterpret_cast<float4*>(p.outF + (size_t)row * p.ldf + idx * 4) = o;
                }
            }
            if (p.outB) {
                if (p.ps_h) {   // PixelShuffle(2) scatter: channel c=4*idx+{0..3} -> (dy,dx) = (j>>1, j&1), out ch idx
                    int y = row / p.ps_w, xx = row - y * p.ps_w;
                    int C4o = p.C >> 2, W2 = 2 * p.ps_w;
                    size_t ob = ((size_t)(2 * y) * W2 + 2 * xx) * C4o + idx;
                    act_store1(p.outB, ob, o.x, p.b32);
                    act_store1(p.outB, ob + C4o, o.y, p.b32);
                    act_store1(p.outB, ob + (size_t)W2 * C4o, o.z, p.b32);
                    act_store1(p.outB, ob + (size_t)W2 * C4o + C4o, o.w, p.b32);
                } else {
                    act_store4(p.outB, (size_t)row * p.ldb + idx * 4, o.x, o.y, o.z, o.w, p.b32);
                }
            }
        }
    }
}

int launch_layernorm(const LnArgs& a, hipStream_t s) {
    UNI_REQUIRE(a.C % 4 == 0 && a.C <= 1536 && a.ldx % 4 == 0, "layernorm: C=%d ldx=%d unsupported", a.C, a.ldx);
    UNI_REQUIRE(((uintptr_t)a.x & 15) == 0, "layernorm: x not 16-B aligned");
    int ni = cdiv(a.C / 4, 64);
    dim3 grid(cdiv(a.M, 4)), block(256);
    switch (ni) {
        case 1: hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, s, a); break;
        case 3: hipLaunchKernelGGL(layernorm_kernel<3>, grid, block, 0, s, a); break;
        case 4: hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL(layernorm_kernel<6>, grid, block, 0, s, a); break;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// GroupNorm apply (stats were accumulated by the producing GEMM's epilogue)
// ------------------------------------------------------------------------------------------------
// ACT: compile-time activation (ACT_NONE / ACT_RELU / ACT_SILU; -1 = the runtime switch of act_apply for anything else).  FAST: the
// reciprocal-based activations of the GEMM epilogues (act_fast) -- used whenever the operand format is not exact fp32.  (One kernel with
// act_apply's runtime switch per element compiled to 5000 instructions and 350 branches: the erff polynomial of a GELU nobody asks for here,
// IEEE division sequences for SiLU.)
template <int ACT, bool FAST>
__global__ __launch_bounds__(256) void gn_apply_kernel(GnApplyArgs p) {
    extern __shared__ float sm[];          // scale[C], shift[C]
    float* scale = sm;
    float* shift = sm + p.C;
    const int cpg = p.C / p.G;
    const int sb = blockIdx.y;                                   // sample of the batch; M = rows PER SAMPLE
    const double* st = p.stats + (size_t)sb * 64;
    const double cnt = (double)p.M * cpg;
    for (int c = threadIdx.x; c < p.C; c += blockDim.x) {
        int g = c / cpg;
        double mean = st[2 * g] / cnt;
        double var = st[2 * g + 1] / cnt - mean * mean;
        float rstd = (float)(1.0 / sqrt((var > 0 ? var : 0) + (double)p.eps));
        float ga = p.gamma[c] * rstd;
        scale[c] = ga;
        shift[c] = p.beta[c] - (float)mean * ga;
    }
    __syncthreads();
    // A thread keeps ONE group of 8 channels for the whole launch: the launcher makes gridDim.x * 256 a multiple of C / 8, so the grid
    // stride moves a thread down the rows of its channel group.  Its 8 scale / shift pairs (and prior betas) live in registers and the
    // row index advances by a constant: no per-item 64-bit divisions, no LDS reads in the loop.
    const unsigned C8 = (unsigned)p.C >> 3;
    const unsigned e0 = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned dml = (gridDim.x * blockDim.x) / C8;         // rows per grid stride
    unsigned ml = e0 / C8;                                       // row within the sample
    const int c = (int)(e0 - ml * C8) * 8;
    const size_t row0 = (size_t)sb * p.M;                        // first global row of this sample
    float sc[8], sh[8], pb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = scale[c + j]; sh[j] = shift[c + j]; pb[j] = p.prior ? p.prior_beta[c + j] : 0.f; }
    auto src_of = [&](unsigned r) __attribute__((always_inline)) {
        return reinterpret_cast<const float4*>(p.x + (row0 + r) * (size_t)p.ldx + c);
    };
    auto process = [&](unsigned r, float4 a, float4 b) __attribute__((always_inline)) {
        const size_t m = row0 + r;
        float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        const float pr = p.prior ? p.prior[m] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float t = v[j] * sc[j] + sh[j];
            float y = ACT < 0 ? act_apply(t, p.act) : (FAST ? act_fast<(ACT < 0 ? 0 : ACT)>(t) : act_apply(t, ACT));
            if (p.prior) y += pr * pb[j];
            v[j] = y;
        }
        if (p.outF) {
            float4* op = reinterpret_cast<float4*>(p.outF + (size_t)m * p.ldf + c);
            op[0] = make_float4(v[0], v[1], v[2], v[3]);
            op[1] = make_float4(v[4], v[5], v[6], v[7]);
        }
        if (p.outB) act_store8(p.outB, (size_t)m * p.ldb + c, v, p.b32);
        if (p.outUp) {
            const unsigned y = r / (unsigned)p.W, x = r - y * (unsigned)p.W;
            size_t W2 = 2 * (size_t)p.W;
            size_t u = (4 * row0 + (size_t)(2 * y) * W2 + 2 * x) * p.ldu + c;
            act_store8(p.outUp, u, v, p.b32);
            act_store8(p.outUp, u + p.ldu, v, p.b32);
            act_store8(p.outUp, u + W2 * p.ldu, v, p.b32);
            act_store8(p.outUp, u + W2 * p.ldu + p.ldu, v, p.b32);
        }
    };
    // two rows per trip with both loads issued first (the stores of one item otherwise fence the loads of the next: the output
    // pointers may alias the input for all the compiler knows)
    const unsigned M = (unsigned)p.M;
    for (; ml + dml < M; ml += 2 * dml) {                        // (four rows per trip measured the same: 4076 vs 4088 GB/s)
        const float4* x0 = src_of(ml);
        const float4* x1 = src_of(ml + dml);
        const float4 a0 = x0[0], b0 = x0[1], a1 = x1[0], b1 = x1[1];
        process(ml, a0, b0);
        process(ml + dml, a1, b1);
    }
    for (; ml < M; ml += dml) {
        const float4* x0 = src_of(ml);
        process(ml, x0[0], x0[1]);
    }
}

int launch_gn_apply(const GnApplyArgs& a, hipStream_t s) {
    UNI_REQUIRE(a.C % 8 == 0 && a.C % a.G == 0 && a.ldx % 4 == 0, "gn_apply: C=%d G=%d ldx=%d", a.C, a.G, a.ldx);
    if (a.outB) UNI_REQUIRE(a.ldb % 8 == 0 && ((uintptr_t)a.outB & 15) == 0, "gn_apply: outB alignment");
    if (a.outUp) UNI_REQUIRE(a.ldu % 8 == 0 && ((uintptr_t)a.outUp & 15) == 0 && a.W > 0, "gn_apply: outUp alignment");
    // every block first derives the C scale / shift pairs from the group sums (fp64 divide + sqrt per channel), so a block should
    // then stream several rows: ~4 float4-pairs per thread (measured best of 1 / 4 / 8 / 16), at least ~8 blocks per CU over the whole batch
    const long total = (long)a.M * (a.C / 8);
    const int nb = a.B > 0 ? a.B : 1;
    static const int per = getenv("UNI_GN_PER") ? atoi(getenv("UNI_GN_PER")) : 4;
    long grid = (total + 256L * per - 1) / (256L * per);
    static const int mg_env = getenv("UNI_GN_MINGRID") ? atoi(getenv("UNI_GN_MINGRID")) : 0;
    // (one frame per call: 1024 instead of 2048 blocks -- every block pays the fp64 prologue -- 0.585 -> 0.552 ms over the 61 launches of a frame)
    const long min_grid = ((mg_env > 0 ? mg_env : (nb == 1 ? 1024 : 2048)) + nb - 1) / nb;
    if (grid < min_grid) grid = min_grid < (total + 255) / 256 ? min_grid : (total + 255) / 256;
    if (grid > 2048) grid = 2048;
    {   // gridDim.x * 256 must be a multiple of C / 8 (a thread keeps its channel group): round the grid up to a multiple of C8 / gcd(C8, 256)
        int c8 = a.C / 8, g = c8, t = 256;
        while (t) { const int r = g % t; g = t; t = r; }
        const int q = c8 / g;
        grid = (grid + q - 1) / q * q;
    }
    const bool fast = a.b32 != FMT_F32;
#define GN_GO(ACT, F) hipLaunchKernelGGL((gn_apply_kernel<ACT, F>), dim3((unsigned)grid, nb), dim3(256), 2 * a.C * sizeof(float), s, a)
    switch (a.act) {
        case ACT_NONE: if (fast) GN_GO(ACT_NONE, true); else GN_GO(ACT_NONE, false); break;
        case ACT_RELU: if (fast) GN_GO(ACT_RELU, true); else GN_GO(ACT_RELU, false); break;
        case ACT_SILU: if (fast) GN_GO(ACT_SILU, true); else GN_GO(ACT_SILU, false); break;
        default: GN_GO(-1, false); break;
    }
#undef GN_GO
    return 0;
}

// ------------------------------------------------------------------------------------------------
// deterministic strip reduction shared by dwconv7_ln and stem:
// thread (sl, cg) holds NV partial sums (its 4 channels) for the NV pixels of strip sl; returns the
// per-pixel totals over all CG channel groups.
// ------------------------------------------------------------------------------------------------
template <int NV>
__device__ __forceinline__ void strip_reduce(const float (&vals)[NV], float (&tot)[NV], float* lds, int S, int CG,
                                             int sl, int cg, bool active) {
    // level 1: every thread parks its NV partials in LDS; level 2: each (strip, pixel) pair is summed by a 16-lane
    // group (strided loads + 4 xor-shuffles) -- deterministic order, ~CG/16 dependent LDS reads instead of CG/8 + CG/8.
    const int T = blockDim.x, tid = threadIdx.x;
    const int NP = S * NV;                 // (strip, pixel) pairs
    float* l1 = lds;                       // [NP][CG]
    float* l3 = l1 + NP * CG;              // [NP]
    if (active) {
#pragma unroll
        for (int o = 0; o < NV; ++o) l1[(sl * NV + o) * CG + cg] = vals[o];
    }
    __syncthreads();
    const int g = tid & 15;
    for (int u = tid >> 4; u < NP; u += T >> 4) {
        float s = 0.f;
        for (int c = g; c < CG; c += 16) s += l1[u * CG + c];
        s += __shfl_xor(s, 8, 64);
        s += __shfl_xor(s, 4, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 1, 64);
        if (g == 0) l3[u] = s;
    }
    __syncthreads();
    if (active) {
#pragma unroll
        for (int o = 0; o < NV; ++o) tot[o] = l3[sl * NV + o];
    }
    __syncthreads();
}
static size_t strip_reduce_lds(int S, int CG, int NV) {
    int NP = S * NV;
    return (size_t)(NP * CG + NP) * sizeof(float);
}

// ------------------------------------------------------------------------------------------------
// depthwise 7x7 + bias + LayerNorm.  thread = (strip of PX pixels along x, 4 channels); a (PX+6)-wide input
// row window is held in registers and reused by the 7 kx taps (register sliding window); the next input row is
// prefetched while the current one is multiplied.  PX = 8 for big maps (fewest loads per output), PX = 4 for small
// maps (2x the waves: the stride-16/32 maps are latency-, not bandwidth-bound).
// ------------------------------------------------------------------------------------------------
template <int IN>
__device__ __forceinline__ void dw_load_row(f32x4 (&dst)[IN], const DwLnArgs& p, size_t img0, int iy, int x0, int cg) {
    const bool rok = iy >= 0 && iy < p.H;
    const float* rowp = p.x + (img0 + (size_t)(rok ? iy : 0) * p.W) * p.C + cg * 4;
#pragma unroll
    for (int j = 0; j < IN; ++j) {
        int ix = x0 + j - 3;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (rok && ix >= 0 && ix < p.W) v = *reinterpret_cast<const f32x4*>(rowp + (size_t)ix * p.C);
        dst[j] = v;
    }
}

template <int IN>
__device__ __forceinline__ void dw_load_row_interior(f32x4 (&dst)[IN], const float* rowp, int C) {
#pragma unroll
    for (int j = 0; j < IN; ++j) dst[j] = *reinterpret_cast<const f32x4*>(rowp + (size_t)j * C);
}

template <int PX>
__global__ __launch_bounds__(512) void dwconv7_ln_kernel(DwLnArgs p, int S, int CG, int spr, int nstrips) {
    constexpr int IN = PX + 6;
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    const int cg = tid % CG, sl = tid / CG;
    // XCD-aware remap (block b runs on XCD b%8, private L2): give each XCD a contiguous band of strips/rows so the
    // 7-row halo re-reads hit its own L2 instead of every XCD pulling the whole map through the fabric.
    int blk;
    {
        const int nwg = gridDim.x, b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int strip = blk * S + sl;
    const bool active = sl < S && strip < nstrips;
    const int yg = active ? strip / spr : 0;                    // row over the whole batch
    const int x0 = active ? (strip - yg * spr) * PX : 0;
    const int sb = yg / p.H, y = yg - sb * p.H;                  // sample, row within the sample
    const size_t img0 = (size_t)sb * p.H * p.W;                  // first pixel of this sample
    const int C = p.C;
    f32x4 acc[PX];
    {
        f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + cg * 4);
#pragma unroll
        for (int o = 0; o < PX; ++o) acc[o] = b;
    }
    if (active) {
        f32x4 cur[IN], nxt[IN];
        // interior strips (all 7 rows and the 6-pixel halo inside the image: > 90 % of a 200x320 map) take a
        // branch-free path; border strips use the predicated loader
        const bool interior = y >= 3 && y + 3 < p.H && x0 >= 3 && x0 + PX + 3 <= p.W;
        if (interior) {
            const float* base = p.x + (img0 + (size_t)(y - 3) * p.W + (x0 - 3)) * C + cg * 4;
            const size_t rstride = (size_t)p.W * C;
            dw_load_row_interior<IN>(cur, base, C);
            // rows 0..5 with the prefetch of the next row, row 6 peeled (a conditional prefetch makes the wait counts conservative: the
            // first FMA then waits for the prefetch it should run beside).  The taps of a row are requested BEFORE the next input row:
            // loads return in order.
            auto taps = [&](f32x4 (&w)[7], int ky) __attribute__((always_inline)) {
                const float* wrow = p.w + (size_t)(ky * 7) * C + cg * 4;
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) w[kx] = *reinterpret_cast<const f32x4*>(wrow + (size_t)kx * C);
            };
            auto mac = [&](const f32x4 (&w)[7]) __attribute__((always_inline)) {
#pragma unroll
                for (int kx = 0; kx < 7; ++kx)
#pragma unroll
                    for (int o = 0; o < PX; ++o)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[o][e] = fmaf(w[kx][e], cur[o + kx][e], acc[o][e]);
            };
#pragma unroll 1
            for (int ky = 0; ky < 6; ++ky) {
                f32x4 w[7];
                taps(w, ky);
                __builtin_amdgcn_sched_barrier(0);
                dw_load_row_interior<IN>(nxt, base + (size_t)(ky + 1) * rstride, C);
                __builtin_amdgcn_sched_barrier(0);
                mac(w);
#pragma unroll
                for (int j = 0; j < IN; ++j) cur[j] = nxt[j];
            }
            {
                f32x4 w[7];
                taps(w, 6);
                mac(w);
            }
        } else {
            dw_load_row<IN>(cur, p, img0, y - 3, x0, cg);
#pragma unroll 1
            for (int ky = 0; ky < 7; ++ky) {
                const float* wrow = p.w + (size_t)(ky * 7) * C + cg * 4;
                f32x4 w[7];
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) w[kx] = *reinterpret_cast<const f32x4*>(wrow + (size_t)kx * C);
                __builtin_amdgcn_sched_barrier(0);
                if (ky < 6) dw_load_row<IN>(nxt, p, img0, y + ky - 2, x0, cg);      // prefetch the next input row
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) {
#pragma unroll
                    for (int o = 0; o < PX; ++o) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[o][e] = fmaf(w[kx][e], cur[o + kx][e], acc[o][e]);
                    }
                }
#pragma unroll
                for (int j = 0; j < IN; ++j) cur[j] = nxt[j];
            }
        }
    }
    float part[PX], tot[PX];
#pragma unroll
    for (int o = 0; o < PX; ++o) part[o] = acc[o][0] + acc[o][1] + acc[o][2] + acc[o][3];
    strip_reduce<PX>(part, tot, lds, S, CG, sl, cg, active);
    float mean[PX];
    const float invC = 1.f / (float)C;
#pragma unroll
    for (int o = 0; o < PX; ++o) {
        mean[o] = tot[o] * invC;
        float a = acc[o][0] - mean[o], b = acc[o][1] - mean[o], c = acc[o][2] - mean[o], d = acc[o][3] - mean[o];
        part[o] = a * a + b * b + c * c + d * d;
    }
    strip_reduce<PX>(part, tot, lds, S, CG, sl, cg, active);
    if (!active) return;
    const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + cg * 4);
    const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + cg * 4);
#pragma unroll
    for (int o = 0; o < PX; ++o) {
        if (x0 + o < p.W) {
            float rstd = rsqrtf(tot[o] * invC + p.eps);
            act_store4(p.out, (img0 + (size_t)y * p.W + x0 + o) * C + cg * 4, (acc[o][0] - mean[o]) * rstd * g[0] + be[0],
                       (acc[o][1] - mean[o]) * rstd * g[1] + be[1], (acc[o][2] - mean[o]) * rstd * g[2] + be[2],
                       (acc[o][3] - mean[o]) * rstd * g[3] + be[3], p.b32);
        }
    }
}

// Two output rows per thread (8 px x 2 rows x 4 ch): the 8 input rows of the window are each loaded once and feed
// both output rows, cutting the L1/TA traffic per output by 1.8x (the single-row kernel is L1-bandwidth bound on
// large maps: ~4.6 16-byte loads per output float4, ~15 TB/s of L1 traffic at 1.3 TB/s algorithmic).
__global__ __launch_bounds__(384) void dwconv7_ln2_kernel(DwLnArgs p, int S, int CG, int spr, int nstrips) {
    constexpr int PX = 8, IN = PX + 6;
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    const int cg = tid % CG, sl = tid / CG;
    int blk;
    {
        const int nwg = gridDim.x, b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int strip = blk * S + sl;
    const bool active = sl < S && strip < nstrips;
    const int HP = (p.H + 1) >> 1;                               // row pairs per image
    const int yg = active ? strip / spr : 0;
    const int x0 = active ? (strip - yg * spr) * PX : 0;
    const int sb = yg / HP, y = (yg - sb * HP) * 2;              // sample, first of the two output rows
    const size_t img0 = (size_t)sb * p.H * p.W;
    const int C = p.C;
    f32x4 acc0[PX], acc1[PX];
    {
        f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + cg * 4);
#pragma unroll
        for (int o = 0; o < PX; ++o) { acc0[o] = b; acc1[o] = b; }
    }
    if (active) {
        f32x4 cur[IN];
#pragma unroll 1
        for (int r = 0; r < 8; ++r) {                            // input row y-3+r: ky = r for row y, ky = r-1 for row y+1
            dw_load_row<IN>(cur, p, img0, y - 3 + r, x0, cg);
            if (r < 7) {
                const float* wrow = p.w + (size_t)(r * 7) * C + cg * 4;
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) {
                    f32x4 w = *reinterpret_cast<const f32x4*>(wrow + (size_t)kx * C);
#pragma unroll
                    for (int o = 0; o < PX; ++o)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc0[o][e] = fmaf(w[e], cur[o + kx][e], acc0[o][e]);
                }
            }
            if (r > 0) {
                const float* wrow = p.w + (size_t)((r - 1) * 7) * C + cg * 4;
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) {
                    f32x4 w = *reinterpret_cast<const f32x4*>(wrow + (size_t)kx * C);
#pragma unroll
                    for (int o = 0; o < PX; ++o)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc1[o][e] = fmaf(w[e], cur[o + kx][e], acc1[o][e]);
                }
            }
        }
    }
    float part[2 * PX], tot[2 * PX], mean[2 * PX];
#pragma unroll
    for (int o = 0; o < PX; ++o) {
        part[o] = acc0[o][0] + acc0[o][1] + acc0[o][2] + acc0[o][3];
        part[PX + o] = acc1[o][0] + acc1[o][1] + acc1[o][2] + acc1[o][3];
    }
    strip_reduce<2 * PX>(part, tot, lds, S, CG, sl, cg, active);
    const float invC = 1.f / (float)C;
#pragma unroll
    for (int o = 0; o < PX; ++o) {
        mean[o] = tot[o] * invC;
        mean[PX + o] = tot[PX + o] * invC;
        float a = acc0[o][0] - mean[o], b = acc0[o][1] - mean[o], c = acc0[o][2] - mean[o], d = acc0[o][3] - mean[o];
        part[o] = a * a + b * b + c * c + d * d;
        a = acc1[o][0] - mean[PX + o]; b = acc1[o][1] - mean[PX + o]; c = acc1[o][2] - mean[PX + o]; d = acc1[o][3] - mean[PX + o];
        part[PX + o] = a * a + b * b + c * c + d * d;
    }
    strip_reduce<2 * PX>(part, tot, lds, S, CG, sl, cg, active);
    if (!active) return;
    const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + cg * 4);
    const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + cg * 4);
#pragma unroll
    for (int o = 0; o < PX; ++o) {
        if (x0 + o < p.W) {
            float rstd = rsqrtf(tot[o] * invC + p.eps);
            act_store4(p.out, (img0 + (size_t)y * p.W + x0 + o) * C + cg * 4, (acc0[o][0] - mean[o]) * rstd * g[0] + be[0],
                       (acc0[o][1] - mean[o]) * rstd * g[1] + be[1], (acc0[o][2] - mean[o]) * rstd * g[2] + be[2],
                       (acc0[o][3] - mean[o]) * rstd * g[3] + be[3], p.b32);
            if (y + 1 < p.H) {
                rstd = rsqrtf(tot[PX + o] * invC + p.eps);
                act_store4(p.out, (img0 + (size_t)(y + 1) * p.W + x0 + o) * C + cg * 4,
                           (acc1[o][0] - mean[PX + o]) * rstd * g[0] + be[0], (acc1[o][1] - mean[PX + o]) * rstd * g[1] + be[1],
                           (acc1[o][2] - mean[PX + o]) * rstd * g[2] + be[2], (acc1[o][3] - mean[PX + o]) * rstd * g[3] + be[3], p.b32);
            }
        }
    }
}

// The same 2-row x 8-px x 4-channel thread as a PERSISTENT block for C = 64 k channels (a wave = 64 channel groups of ONE strip, so
// everything about the strip is wave-uniform):
//   * input rows through a BUFFER DESCRIPTOR per row (base = row start, num_records = row bytes or 0 for a row outside the
//     image) and 14 per-lane offsets computed once per strip: a pixel left / right of the row is an offset outside the
//     descriptor and reads zeros -- no bounds branches (the kernel above spends ~45 branches per row on them), no 64-bit
//     address math, and few enough registers for
//   * a register double buffer: the next input row is in flight while this one is multiplied (the kernel above exposes the L2
//     round trip of every row: 2 waves per SIMD cannot cover it);
//   * the 49 x C weights resident in LDS (conflict-free 16-byte reads) instead of 6 weight float4s per output float4 through the
//     vector L1, which a 150 KB table (C = 768) does not fit;
//   * LayerNorm sums reduced inside the waves (reduce-scatter butterfly: 17 shuffles for the 16 pixels), 64 B of scratch per wave.
// ROWS = 4 output rows per thread: the measured bound of these kernels is the L2 -> CU load path (~22-25 B/clk/CU, the same figure
// the GEMM's LDS-DMA sees), and 10 input rows for 4 output rows is 4.4 input float4s per output float4 instead of 7.
// DBG (tools/dwln_bench.py with UNI_DW_DBG, C = 768 / ROWS = 2 only): ablation builds -- 1 no input loads, 2 no LayerNorm reductions
// (local statistics), 4 no output stores, 8 no FMAs.  DBG = 0 is the kernel the engine runs.
// NW = 12 (ROWS = 2, round 4): twelve waves per block = THREE per SIMD (<= 168 registers: one 7-tap buffer re-read in place instead of the tap double
// buffer).  A PMC pass showed the 8-wave kernel at C = 768 as 6 waves on 4 SIMDs (two SIMDs 52 % VALU-busy, two 26 %) with 43 % of the wave cycles waiting.
// PACK (C = 384 / 192, round 4): CG = 96 / 48 channel groups leave a quarter of the lanes of a strip's waves idle.  Packed, TWO (C = 384) or FOUR
// (C = 192) strips side by side in x -- a "strip group" of 16 / 32 px -- fill exactly three waves: flat lane f of the group is channel group f % CG of
// strip f / CG.  Everything about the ROW stays wave-uniform; the strip of a lane is 8 px * (f / CG) further right, which travels in the per-lane
// buffer offset (descriptor base AT the first pixel, so the range check drops whatever lies past the row end; at the left image edge the base sits
// on the second strip's pixel and the first strip's lanes carry a negative = out-of-range offset).  The LayerNorm sums are reduce-scattered inside
// 32- / 16-lane granules (three granules per strip, like the three waves of a C = 768 strip).  S / spr / nstrips then count strip GROUPS.
template <int C, int ROWS, int DBG = 0, int NW = 8, bool PACK = false>            // C compile-time: the tap offsets become instruction immediates
__global__ __launch_bounds__(NW * 64) void dwconv7_lnb_kernel(DwLnArgs p, int S, int spr, int nstrips) {
    constexpr int PX = 8, IN = PX + 6, CG = C / 4;
    static_assert(!PACK || CG == 96 || CG == 48, "packed lanes: C = 384 / 192");
    constexpr int G = !PACK ? 64 : (CG % 32 == 0 ? 32 : 16);    // lanes of a reduction granule
    constexpr int LG = G == 64 ? 2 : G == 32 ? 1 : 0;            // lane bits below the four reduce-scatter bits
    constexpr int NSLOT = NW * 64 / G;                           // granules of a block
    extern __shared__ float lds[];
    float* wl = lds;                                             // [49][C]
    float* red = lds + 49 * C;                                   // [2][waves][16]
    const int tid = threadIdx.x;
    for (int u = tid; u < 49 * C / 4; u += blockDim.x) reinterpret_cast<f32x4*>(wl)[u] = reinterpret_cast<const f32x4*>(p.w)[u];
    __syncthreads();
    constexpr int wps = PACK ? 3 : (CG + 63) >> 6;               // waves per strip (group); unpacked, C = 192 / 384 leave the lanes past CG idle
    constexpr int NGS = PACK ? CG / G : wps;                     // granules per strip
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sl = wv / wps, fl = (wv - sl * wps) * 64 + lane;
    const int ks = PACK ? fl / CG : 0;                           // strip of the group (per lane)
    const int cg0 = fl - ks * CG;
    const bool lane_ok = PACK || cg0 < CG;
    const int cg = lane_ok ? cg0 : 0;                            // idle lanes shadow channel group 0 and store nothing
    const int kso = ks * PX * C;                                 // the lane's strip in elements from the group's first pixel
    int first, step, count;
    {
        const int nblk = (nstrips + S - 1) / S;
        const int x = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
        const int q = nblk >> 3, r = nblk & 7;
        const int start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
        const int cnt = q + (x < r ? 1 : 0);
        first = start + slot;
        step = nslots;
        count = slot < cnt ? (cnt - slot + nslots - 1) / nslots : 0;
    }
    const int HP = (p.H + ROWS - 1) / ROWS;                      // row groups per image
    constexpr float invC = 1.f / (float)C;
    const int rowbytes = p.W * C * 4;
    // sum over the lanes of a G-lane granule (G = 64: the wave) of 16 values at once: afterwards every lane holds the granule total of value
    // (lane >> LG) & 15
    auto wave_scatter_sum = [&](const float (&v)[16]) __attribute__((always_inline)) {
        float a8[8], a4[4], a2[2], a1;
        {
            const bool up = (lane >> (LG + 3)) & 1;
#pragma unroll
            for (int r = 0; r < 8; ++r) a8[r] = (up ? v[r + 8] : v[r]) + __shfl_xor(up ? v[r] : v[r + 8], 8 << LG, 64);
        }
        {
            const bool up = (lane >> (LG + 2)) & 1;
#pragma unroll
            for (int r = 0; r < 4; ++r) a4[r] = (up ? a8[r + 4] : a8[r]) + __shfl_xor(up ? a8[r] : a8[r + 4], 4 << LG, 64);
        }
        {
            const bool up = (lane >> (LG + 1)) & 1;
#pragma unroll
            for (int r = 0; r < 2; ++r) a2[r] = (up ? a4[r + 2] : a4[r]) + __shfl_xor(up ? a4[r] : a4[r + 2], 2 << LG, 64);
        }
        {
            const bool up = (lane >> LG) & 1;
            a1 = (up ? a2[1] : a2[0]) + __shfl_xor(up ? a2[0] : a2[1], 1 << LG, 64);
        }
        if (LG >= 2) a1 += __shfl_xor(a1, 2, 64);
        if (LG >= 1) a1 += __shfl_xor(a1, 1, 64);
        return a1;
    };
    // `buf` alternates between the two scratch buffers from one reduction to the next: ONE barrier per reduction is enough (a wave that writes
    // buffer b again has passed the barrier of the reduction in between, which every wave reaches only after its reads of b)
    auto reduce16 = [&](const float (&part)[16], float (&tot)[16], int buf) __attribute__((always_inline)) {
        const float t = wave_scatter_sum(part);
        float* rb = red + buf * (NSLOT * 16);
        if ((lane & ((1 << LG) - 1)) == 0) rb[((wv * 64 + lane) / G) * 16 + ((lane >> LG) & 15)] = t;
        // a strip that is ONE wave (C <= 256) needs no block barrier: the strips of a block are independent
        if (wps == 1) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
        else __syncthreads();
#pragma unroll
        for (int o = 0; o < 16; ++o) tot[o] = 0.f;
        const int slot0 = (sl * wps * 64 + ks * CG) / G;           // first granule of the lane's strip
#pragma unroll
        for (int w_ = 0; w_ < NGS; ++w_) {                       // fixed order: deterministic; four 16-byte broadcast reads per granule of the strip
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 v4 = *reinterpret_cast<const f32x4*>(rb + (slot0 + w_) * 16 + 4 * g4);
#pragma unroll
                for (int e = 0; e < 4; ++e) tot[4 * g4 + e] += v4[e];
            }
        }
    };
#pragma unroll 1
    for (int it = 0; it < count; ++it) {
        asm volatile("" ::: "memory");                           // keep the 49 weight reads inside the row loop (hoisted out of this loop they cost 196 registers)
        const int strip = (first + it * step) * S + sl;          // wave-uniform
        const bool active = sl < S && strip < nstrips;
        const int yg = active ? strip / spr : 0;
        const int x0 = active ? (strip - yg * spr) * PX * (PACK ? 192 / CG : 1) : 0;
        const int sb = yg / HP, y = (yg - sb * HP) * ROWS;       // sample, first of the output rows
        const size_t img0 = (size_t)sb * p.H * p.W;
        f32x4 acc[ROWS][PX];
        {
            const f32x4 bias4 = *reinterpret_cast<const f32x4*>(p.bias + cg * 4);
#pragma unroll
            for (int q = 0; q < ROWS; ++q)
#pragma unroll
                for (int o = 0; o < PX; ++o) acc[q][o] = bias4;
        }
        if (active) {
            // one per-lane offset (the channel group) for all loads; the pixel travels in the scalar offset and a pixel outside the
            // row (wave-uniform) or a row outside the image gets num_records = 0: the load returns zeros
            auto load_row = [&](f32x4 (&dst)[IN], int iy) __attribute__((always_inline)) {
                const bool rok = iy >= 0 && iy < p.H;
                float* rowp = const_cast<float*>(p.x) + (img0 + (size_t)(rok ? iy : 0) * p.W) * C;
#pragma unroll
                for (int j = 0; j < IN; ++j) {
                    const int ix = x0 + j - 3;
                    if constexpr (PACK) {
                        // descriptor base AT the pixel (the range check covers the per-lane offset only); left of the image (j < 3 of the first
                        // group of a row) the base is the SECOND strip's pixel and the first strip's lanes carry a negative offset: out of range
                        const bool neg = j < 3 && ix < 0;
                        const int ixb = neg ? ix + PX : ix;
                        const int rec = rok ? max(p.W - ixb, 0) * C * 4 : 0;
                        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(rowp + (size_t)max(ixb, 0) * C, 0, rec, 0x00020000);
                        const int vo = cg * 16 + kso * 4;
                        dst[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, neg ? vo - PX * C * 4 : vo, 0, 0));
                    } else {
                        const bool ok = rok && ix >= 0 && ix < p.W;
                        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(rowp, 0, ok ? rowbytes : 0, 0x00020000);
                        dst[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, cg * 16, ok ? ix * C * 4 : 0, 0));
                    }
                }
            };
            // Weights: the 7 taps of ONE (input row, output row) pair at a time through a register double buffer -- the reads of the
            // next pair (from LDS, ~100+ cycles) are issued before the 224 FMAs of this one.  (Letting the compiler hoist them
            // freely costs 100+ registers; fencing every pair with sched_barrier left the waves parked on lgkmcnt ~50 % of their
            // cycles.)
            auto ld_w = [&](f32x4 (&w)[7], int ky) __attribute__((always_inline)) {
                const float* wrow = wl + (min(max(ky, 0), 6) * 7) * C + cg * 4;
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) w[kx] = *reinterpret_cast<const f32x4*>(wrow + kx * C);
            };
            auto mac = [&](f32x4 (&a)[PX], const f32x4 (&w)[7], const f32x4 (&src)[IN]) __attribute__((always_inline)) {
#pragma unroll
                for (int kx = 0; kx < 7; ++kx)
#pragma unroll
                    for (int o = 0; o < PX; ++o)
#pragma unroll
                        for (int e = 0; e < 4; ++e) a[o][e] = fmaf(w[kx][e], src[o + kx][e], a[o][e]);
            };
            if constexpr (NW > 8) {
                // one 7-tap buffer: each tap is re-read IN PLACE for the next (input row, output row) pair right after its last use (one tap = 16
                // packed FMAs = 64 cycles, the read has six taps of slack).  Pair sequence (r, q) = (0,0) (0,1) (1,0) (1,1) ...: tap row ky = r - q;
                // the pair after (r, 0) is (r, 1) with tap row r - 1, the pair after (r, 1) is (r + 1, 0) with tap row r + 1 (clamped: out-of-range
                // pairs multiply nothing and only pass the taps on)
                f32x4 ra[IN], w[7];
                {
                    const float* wrow = wl + cg * 4;
#pragma unroll
                    for (int kx = 0; kx < 7; ++kx) w[kx] = *reinterpret_cast<const f32x4*>(wrow + kx * C);
                }
                auto pair = [&](f32x4 (&a)[PX], int ky, int nk) __attribute__((always_inline)) {
                    const float* wn = wl + (min(max(nk, 0), 6) * 7) * C + cg * 4;
                    if (!(DBG & 8) && ky >= 0 && ky < 7) {           // wave-uniform
#pragma unroll
                        for (int kx = 0; kx < 7; ++kx) {
#pragma unroll
                            for (int o = 0; o < PX; ++o)
#pragma unroll
                                for (int e = 0; e < 4; ++e) a[o][e] = fmaf(w[kx][e], ra[o + kx][e], a[o][e]);
                            w[kx] = *reinterpret_cast<const f32x4*>(wn + kx * C);
                        }
                    } else {
#pragma unroll
                        for (int kx = 0; kx < 7; ++kx) w[kx] = *reinterpret_cast<const f32x4*>(wn + kx * C);
                    }
                };
#pragma unroll 1
                for (int r = 0; r < 8; ++r) {
                    load_row(ra, y - 3 + r);
                    pair(acc[0], r, r - 1);
                    pair(acc[1], r - 1, r + 1);
                }
            } else {
            f32x4 ra[IN], wa[7], wb[7];
            ld_w(wa, 0);                                         // pair (r = 0, q = 0): tap row 0
            if (DBG & 1) {
#pragma unroll
                for (int j = 0; j < IN; ++j) ra[j] = f32x4{1.f, 2.f, 3.f, 4.f};
            }
#pragma unroll 1
            for (int r = 0; r < 6 + ROWS; ++r) {                 // input row y-3+r feeds output row y+q with tap row ky = r - q
                if (!(DBG & 1)) load_row(ra, y - 3 + r);
#pragma unroll
                for (int q = 0; q < ROWS; q += 2) {
                    ld_w(wb, r - (q + 1));                       // next pair: (r, q + 1)
                    __builtin_amdgcn_sched_barrier(0);
                    if (!(DBG & 8) && r - q >= 0 && r - q < 7) mac(acc[q], wa, ra);
                    __builtin_amdgcn_sched_barrier(0);
                    ld_w(wa, q + 2 < ROWS ? r - (q + 2) : r + 1); // (r, q + 2), or (r + 1, 0)
                    __builtin_amdgcn_sched_barrier(0);
                    if (!(DBG & 8) && r - q - 1 >= 0 && r - q - 1 < 7) mac(acc[q + 1], wb, ra);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            }
        }
        // LayerNorm per pixel: two-pass (mean, then centred variance), 16 pixels per reduction
        float mean[ROWS][PX], rstd[ROWS][PX];
#pragma unroll
        for (int h = 0; h < ROWS / 2; ++h) {
            float part[16], tot[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const f32x4 v = acc[2 * h + (u >> 3)][u & 7];
                part[u] = lane_ok ? v[0] + v[1] + v[2] + v[3] : 0.f;
            }
            if (DBG & 2) { for (int u = 0; u < 16; ++u) tot[u] = part[u] * (float)CG; } else reduce16(part, tot, 0);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const float m = tot[u] * invC;
                mean[2 * h + (u >> 3)][u & 7] = m;
                const f32x4 v = acc[2 * h + (u >> 3)][u & 7];
                const float a = v[0] - m, b = v[1] - m, c = v[2] - m, d = v[3] - m;
                part[u] = lane_ok ? a * a + b * b + c * c + d * d : 0.f;
            }
            if (DBG & 2) { for (int u = 0; u < 16; ++u) tot[u] = part[u] * (float)CG; } else reduce16(part, tot, 1);
#pragma unroll
            for (int u = 0; u < 16; ++u) rstd[2 * h + (u >> 3)][u & 7] = rsqrtf(tot[u] * invC + p.eps);
        }
        if (active && lane_ok && !((DBG & 4) && p.eps > 0.f)) {       // (DBG 4: the results stay live through the runtime eps test, nothing is stored)
            // 16-byte stores: a lane owns 4 channels = 8 bytes of hi + 8 of lo (f16x2) or 8 bytes of bf16, and 8-byte stores are
            // store-issue bound (32 per thread and tile).  Lane pairs (cg, cg ^ 1; CG is even) swap halves through DPP: f16x2 --
            // the even lane writes the 16 hi bytes of the 8-channel group, the odd lane the 16 lo bytes; bf16 -- the even lane writes
            // the group of pixel o, the odd lane that of pixel o + 1.  A wave instruction then covers 1 KiB of contiguous output.
            const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + cg * 4);
            const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + cg * 4);
            const bool odd = cg & 1;
            auto xor1 = [](unsigned v) __attribute__((always_inline)) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true); };
            auto norm4 = [&](int q, int o, float (&yv)[4]) __attribute__((always_inline)) {
                const float m = mean[q][o], rs = rstd[q][o];
#pragma unroll
                for (int e = 0; e < 4; ++e) yv[e] = (acc[q][o][e] - m) * rs * g[e] + be[e];
            };
#pragma unroll
            for (int q = 0; q < ROWS; ++q) {
                if (y + q >= p.H) continue;
                // one buffer descriptor per output PIXEL (scalar base + ONE per-lane offset: no 64-bit vector addresses -- the 12-wave build spilled
                // and re-loaded them in front of every store): num_records = the bytes from this pixel to the end of the row, so a pixel past the
                // row end (and the odd lane's pixel o + 1 of the bf16 pairing) is dropped by the range check.  (The check covers the per-lane
                // offset only, NOT a scalar offset: the pixel must sit in the base.)
                const int eb = p.b32 == FMT_BF16 ? 2 : 4;
                char* orow = reinterpret_cast<char*>(p.out) + (img0 + (size_t)(y + q) * p.W + x0) * C * eb;
                auto odesc = [&](int o) __attribute__((always_inline)) {
                    return __builtin_amdgcn_make_buffer_rsrc(orow + (size_t)o * C * eb, 0, max(p.W - x0 - o, 0) * C * eb, 0x00020000);
                };
                if (p.b32 == FMT_H2) {
#pragma unroll
                    for (int o = 0; o < PX; ++o) {
                        float yv[4];
                        norm4(q, o, yv);
                        f16x4 h, l;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { f16 hh, ll; h2_split(yv[e], hh, ll); h[e] = hh; l[e] = ll; }
                        const u32x2 hu = __builtin_bit_cast(u32x2, h), lu = __builtin_bit_cast(u32x2, l);
                        const unsigned s0 = odd ? hu[0] : lu[0], s1 = odd ? hu[1] : lu[1];
                        const unsigned r0 = xor1(s0), r1 = xor1(s1);
                        const u32x4 ov = odd ? u32x4{r0, r1, lu[0], lu[1]} : u32x4{hu[0], hu[1], r0, r1};
                        __builtin_amdgcn_raw_buffer_store_b128(ov, odesc(o), (cg & ~1) * 16 + (odd ? 16 : 0) + kso * 4, 0, 0);
                    }
                } else if (p.b32 == FMT_BF16) {
#pragma unroll
                    for (int o = 0; o < PX; o += 2) {
                        float y0[4], y1[4];
                        norm4(q, o, y0);
                        norm4(q, o + 1, y1);
                        const bf16x4 b0 = {(bf16)y0[0], (bf16)y0[1], (bf16)y0[2], (bf16)y0[3]}, b1 = {(bf16)y1[0], (bf16)y1[1], (bf16)y1[2], (bf16)y1[3]};
                        const u32x2 u0 = __builtin_bit_cast(u32x2, b0), u1 = __builtin_bit_cast(u32x2, b1);
                        const unsigned s0 = odd ? u0[0] : u1[0], s1 = odd ? u0[1] : u1[1];
                        const unsigned r0 = xor1(s0), r1 = xor1(s1);
                        const u32x4 ov = odd ? u32x4{r0, r1, u1[0], u1[1]} : u32x4{u0[0], u0[1], r0, r1};
                        __builtin_amdgcn_raw_buffer_store_b128(ov, odesc(o), (cg & ~1) * 8 + (odd ? C * 2 : 0) + kso * 2, 0, 0);      // the odd lane's pixel o + 1 sits in its lane offset
                    }
                } else {
#pragma unroll
                    for (int o = 0; o < PX; ++o) {
                        float yv[4];
                        norm4(q, o, yv);
                        const f32x4 yo = {yv[0], yv[1], yv[2], yv[3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, yo), odesc(o), cg * 16 + kso * 4, 0, 0);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// ONE FRAME PER CALL (round 6): depthwise 7x7 + LayerNorm on a map with too few strips for the persistent kernel above (50 x 80 x 768 of one
// frame = 250 strips of 2 rows x 8 px).  With ~1 wave per SIMD nothing hides the L2 round trip of an input row behind another wave, and a thread
// that walks the 7 (8) rows of its window pays them one after the other (dwconv7_ln_kernel<8>: 18.7-20.8 us per launch, 27 launches per frame;
// the LDS-tap kernel with one strip per block, one wave per SIMD: 26.9 us -- measured, round 6).  Here the 8 input rows of a strip are SPLIT over
// FOUR WAVE GROUPS of one block (group g takes rows 2g, 2g + 1 for both output rows: 3 / 4 / 4 / 3 tap rows), so a wave's dependent chain is two
// rows long and a CU holds 12 waves (three per SIMD) of the same strip; groups 1-3 park their partial sums in LDS ([group][pixel][lane] float4:
// conflict-free 16-byte accesses), group 0 adds them in a fixed order (deterministic), runs the two-pass LayerNorm (in-wave reduce-scatter
// butterfly + one LDS exchange between its waves) and stores.  Taps come straight from global memory (L2-resident 49 x C table, read once per
// block like every other variant).  C = 768 (12 waves); <= 168 registers.
// ------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(C) void dwconv7_lns_kernel(DwLnArgs p, int spr, int nstrips) {
    constexpr int PX = 8, IN = PX + 6, CG = C / 4, WPG = CG / 64;      // waves per group
    static_assert(CG % 64 == 0 && 4 * WPG <= 16, "row-split kernel: C a multiple of 256, at most 16 waves");
    extern __shared__ float lds[];
    f32x4* part = reinterpret_cast<f32x4*>(lds);                       // [3][16][CG]
    float* red = lds + 3 * 16 * CG * 4;                                // [2][WPG][16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int g = __builtin_amdgcn_readfirstlane(tid / CG);            // row group (wave-uniform: CG is a multiple of 64)
    const int cg = tid - g * CG, wv = cg >> 6;
    int blk;
    {
        const int nwg = gridDim.x, b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    if (blk >= nstrips) return;                                        // (whole block)
    const int HP = (p.H + 1) >> 1;
    const int yg = blk / spr, x0 = (blk - yg * spr) * PX;
    const int sb = yg / HP, y = (yg - sb * HP) * 2;
    const size_t img0 = (size_t)sb * p.H * p.W;
    const int rowbytes = p.W * C * 4;
    f32x4 acc[2][PX];
    {
        const f32x4 bias4 = *reinterpret_cast<const f32x4*>(p.bias + cg * 4);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int o = 0; o < PX; ++o) acc[q][o] = g == 0 ? bias4 : z;
    }
    auto mac = [&](f32x4 (&a)[PX], const f32x4 (&w)[7], const f32x4 (&src)[IN]) __attribute__((always_inline)) {
#pragma unroll
        for (int kx = 0; kx < 7; ++kx)
#pragma unroll
            for (int o = 0; o < PX; ++o)
#pragma unroll
                for (int e = 0; e < 4; ++e) a[o][e] = fmaf(w[kx][e], src[o + kx][e], a[o][e]);
    };
    auto taps = [&](f32x4 (&w)[7], int ky) __attribute__((always_inline)) {
        const float* wrow = p.w + (size_t)(ky * 7) * C + cg * 4;
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) w[kx] = *reinterpret_cast<const f32x4*>(wrow + (size_t)kx * C);
    };
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int r = 2 * g + rr, iy = y - 3 + r;                      // window row r: tap row r for output row y, r - 1 for y + 1
        f32x4 row[IN];
        {
            const bool rok = iy >= 0 && iy < p.H;
            float* rowp = const_cast<float*>(p.x) + (img0 + (size_t)(rok ? iy : 0) * p.W) * C;
#pragma unroll
            for (int j = 0; j < IN; ++j) {                             // zero padding = a descriptor with no records (wave-uniform)
                const int ix = x0 + j - 3;
                const bool ok = rok && ix >= 0 && ix < p.W;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(rowp, 0, ok ? rowbytes : 0, 0x00020000);
                row[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, cg * 16, ok ? ix * C * 4 : 0, 0));
            }
        }
        if (r <= 6) {
            f32x4 w[7];
            taps(w, r);
            mac(acc[0], w, row);
        }
        if (r >= 1) {
            f32x4 w[7];
            taps(w, r - 1);
            mac(acc[1], w, row);
        }
    }
    // partial sums of groups 1..3 -> LDS, group 0 adds them in group order
    if (g > 0) {
#pragma unroll
        for (int u = 0; u < 16; ++u) part[((g - 1) * 16 + u) * CG + cg] = acc[u >> 3][u & 7];
    }
    __syncthreads();
    if (g == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const f32x4 v = part[(k * 16 + u) * CG + cg];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[u >> 3][u & 7][e] += v[e];
            }
    }
    // LayerNorm over the C channels of each of the 16 pixels (group 0 computes; every wave takes the barriers)
    auto wave_scatter_sum = [&](const float (&v)[16]) __attribute__((always_inline)) {       // -> lane holds the wave total of value (lane >> 2) & 15
        float a8[8], a4[4], a2[2], a1;
        { const bool up = (lane >> 5) & 1;
#pragma unroll
          for (int r = 0; r < 8; ++r) a8[r] = (up ? v[r + 8] : v[r]) + __shfl_xor(up ? v[r] : v[r + 8], 32, 64); }
        { const bool up = (lane >> 4) & 1;
#pragma unroll
          for (int r = 0; r < 4; ++r) a4[r] = (up ? a8[r + 4] : a8[r]) + __shfl_xor(up ? a8[r] : a8[r + 4], 16, 64); }
        { const bool up = (lane >> 3) & 1;
#pragma unroll
          for (int r = 0; r < 2; ++r) a2[r] = (up ? a4[r + 2] : a4[r]) + __shfl_xor(up ? a4[r] : a4[r + 2], 8, 64); }
        { const bool up = (lane >> 2) & 1;
          a1 = (up ? a2[1] : a2[0]) + __shfl_xor(up ? a2[0] : a2[1], 4, 64); }
        a1 += __shfl_xor(a1, 2, 64);
        a1 += __shfl_xor(a1, 1, 64);
        return a1;
    };
    auto reduce16 = [&](const float (&pt)[16], float (&tot)[16], int buf) __attribute__((always_inline)) {
        float* rb = red + buf * (WPG * 16);
        if (g == 0) {
            const float t = wave_scatter_sum(pt);
            if ((lane & 3) == 0) rb[wv * 16 + ((lane >> 2) & 15)] = t;
        }
        __syncthreads();
#pragma unroll
        for (int o = 0; o < 16; ++o) tot[o] = 0.f;
#pragma unroll
        for (int w_ = 0; w_ < WPG; ++w_)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 v4 = *reinterpret_cast<const f32x4*>(rb + w_ * 16 + 4 * g4);
#pragma unroll
                for (int e = 0; e < 4; ++e) tot[4 * g4 + e] += v4[e];
            }
    };
    constexpr float invC = 1.f / (float)C;
    float pt[16], tot[16], mean[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { const f32x4 v = acc[u >> 3][u & 7]; pt[u] = v[0] + v[1] + v[2] + v[3]; }
    reduce16(pt, tot, 0);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        mean[u] = tot[u] * invC;
        const f32x4 v = acc[u >> 3][u & 7];
        const float a = v[0] - mean[u], b = v[1] - mean[u], c = v[2] - mean[u], d = v[3] - mean[u];
        pt[u] = a * a + b * b + c * c + d * d;
    }
    reduce16(pt, tot, 1);
    if (g != 0) return;
    const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma + cg * 4);
    const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + cg * 4);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int q = u >> 3, o = u & 7;
        if (y + q < p.H && x0 + o < p.W) {
            const float rstd = rsqrtf(tot[u] * invC + p.eps), m = mean[u];
            const f32x4 v = acc[q][o];
            act_store4(p.out, (img0 + (size_t)(y + q) * p.W + x0 + o) * C + cg * 4, (v[0] - m) * rstd * gm[0] + be[0], (v[1] - m) * rstd * gm[1] + be[1],
                       (v[2] - m) * rstd * gm[2] + be[2], (v[3] - m) * rstd * gm[3] + be[3], p.b32);
        }
    }
}

int launch_dwconv7_ln(const DwLnArgs& a, hipStream_t s) {
    UNI_REQUIRE(a.C % 4 == 0 && a.C / 4 <= 512, "dwconv7_ln: C=%d unsupported", a.C);
    const int CG = a.C / 4;
    int S = 256 / CG;
    if (S < 1) S = 1;
    const int T = cdiv(S * CG, 64) * 64;
    // variant by the thread count n8 of the 8-px single-row tiling (measured crossovers, tools/norm_bench.py):
    // n8 >= 150k -> 2 rows x 8 px, >= 70k -> 8 px, else 4 px (more, smaller threads to fill the 256 CUs)
    static const char* env = getenv("UNI_DW_PX");
    const int nb = a.B > 0 ? a.B : 1;
    const long n8 = (long)cdiv(a.W, 8) * a.H * nb * CG;
    int px = n8 >= 150000L ? 16 : n8 >= 70000L ? 8 : 4;
    if (env) px = atoi(env);
    // one frame per call, C = 768: the row-split kernel (above) while the map is at most ONE round of one-strip blocks; UNI_DW_SPLIT=0 = A/B switch
    static const int split_env = getenv("UNI_DW_SPLIT") ? atoi(getenv("UNI_DW_SPLIT")) : 1;
    if (split_env && !env && a.C == 768 && (long)a.W * a.C * 4 < (1L << 30)) {
        const int spr = cdiv(a.W, 8), nst = spr * cdiv(a.H, 2) * nb;
        if (nst <= 256 || split_env == 2) {      // one round of blocks (242 vs 148 us at 16 frames, 31 us at two)      // (UNI_DW_SPLIT=2: every map -- the 16-frame experiment of profiles/r06_dwconv_b1_rowsplit.txt)
            constexpr int ldsb = 3 * 16 * 192 * 16 + 2 * 3 * 16 * 4;
            static DevOnce once;
            UNI_LDS_OPTIN(once, "dwconv7_lns", ldsb, reinterpret_cast<const void*>(&dwconv7_lns_kernel<768>));
            hipLaunchKernelGGL((dwconv7_lns_kernel<768>), dim3(nst), dim3(768), ldsb, s, a, spr, nst);
            return 0;
        }
    }
    if (px == 16) {
        const int spr = cdiv(a.W, 8), nstrips = spr * ((a.H + 1) / 2) * nb;
        // persistent variant (row descriptors, register double buffer, weights in LDS): C a multiple of 256, 49 C floats + scratch
        // within 160 KB, at least two rounds of work for 256 CUs and a row shorter than 2 GiB / 4
        static const bool no_b = getenv("UNI_DW_NOLDSW") != nullptr;
        if (!no_b && (long)a.W * a.C * 4 < (1L << 30) && (a.C == 192 || a.C == 256 || a.C == 384 || a.C == 512 || a.C == 768)) {
            // output rows per thread: 4 (4.4 input float4s per output float4 instead of 7) where the 128 accumulators leave the
            // other registers unspilled (C = 256 / 512: one wave per strip), else 2 (tools/dwln_bench.py)
            static const int rows_env = getenv("UNI_DW_ROWS") ? atoi(getenv("UNI_DW_ROWS")) : 0;
            int rows = rows_env == 2 || rows_env == 4 ? rows_env : (a.C == 256 || a.C == 512 ? 4 : 2);
            const int wps = cdiv(CG, 64), Sw = 8 / wps;          // 512 threads: 8 waves
            const size_t ldsw = (size_t)49 * a.C * 4 + (size_t)2 * 8 * 16 * 4;
            // a map with too few 4-row strips for 1.5 rounds of blocks (the 50 x 80 head level at 16 frames) still fills them with 2-row strips
            if (!rows_env && rows == 4 && cdiv(spr * cdiv(a.H, 4) * nb, Sw) < 384) rows = 2;
            const int nst = spr * cdiv(a.H, rows) * nb;
            if (cdiv(nst, Sw) >= 384) {
                static DevOnce attr_once;
#define DWB_ALL(F) F(192, 2) F(256, 2) F(384, 2) F(512, 2) F(768, 2) F(192, 4) F(256, 4) F(384, 4) F(512, 4) F(768, 4)
#define DWB_ATTR(CC, RR) reinterpret_cast<const void*>(&dwconv7_lnb_kernel<CC, RR>),
                UNI_LDS_OPTIN(attr_once, "dwconv7_lnb", 163840, DWB_ALL(DWB_ATTR) reinterpret_cast<const void*>(&dwconv7_lnb_kernel<768, 2>));
#undef DWB_ATTR
                const dim3 grid(256), block(Sw * wps * 64);
                static const int dbg = getenv("UNI_DW_DBG") ? atoi(getenv("UNI_DW_DBG")) : 0;      // ablation builds of the stage-2 kernel (tools/dwln_bench.py)
                if (dbg && a.C == 768 && rows == 2) {
#define DWB_DBG(D) if (dbg == D) { static DevOnce once; UNI_LDS_OPTIN(once, "dwconv7_lnb (ablation)", 163840, reinterpret_cast<const void*>(&dwconv7_lnb_kernel<768, 2, D>)); \
                                   hipLaunchKernelGGL((dwconv7_lnb_kernel<768, 2, D>), grid, block, ldsw, s, a, Sw, spr, nst); return 0; }
                    DWB_DBG(1) DWB_DBG(2) DWB_DBG(4) DWB_DBG(8) DWB_DBG(3) DWB_DBG(9) DWB_DBG(15)
#undef DWB_DBG
                }
                // packed lanes for C = 384 / 192 (two / four strips side by side fill three waves; see the kernel's header).  UNI_DW_PACK = 0 / 1 (A/B
                // switch), UNI_DW_PACK_NW = 6 / 9 / 12 waves per block (default per C below)
                static const int pack = getenv("UNI_DW_PACK") ? atoi(getenv("UNI_DW_PACK")) : 1;
                if (pack && rows == 2 && (a.C == 384 || a.C == 192)) {
                    static const int nw_env = getenv("UNI_DW_PACK_NW") ? atoi(getenv("UNI_DW_PACK_NW")) : 0;
                    const int sps = 192 / CG, sprg = cdiv(spr, sps), nstg = sprg * cdiv(a.H, rows) * nb;
                    // 12 waves (four groups per block: 602 -> 528 us / 298 -> 265 us on the stage-0 / stage-1 maps at 16 frames; 9 waves 559 / 276, 6 waves
                    // 579 / 292).  One frame of the stage-0 map has too few groups for 12-wave blocks and measured 1 % slower on 6-wave blocks than unpacked.
                    const int nw = nw_env == 6 || nw_env == 9 || nw_env == 12 ? nw_env : 12;
                    const int Sg = nw / 3;
                    const int gran = a.C == 384 ? 32 : 16;
                    const size_t ldsp = (size_t)49 * a.C * 4 + (size_t)2 * (nw * 64 / gran) * 16 * 4;
                    if (cdiv(nstg, Sg) >= 384) {
#define DWB_PACK(CC, NWW) if (a.C == CC && nw == NWW) { static DevOnce once; UNI_LDS_OPTIN(once, "dwconv7_lnb (packed lanes)", 163840, reinterpret_cast<const void*>(&dwconv7_lnb_kernel<CC, 2, 0, NWW, true>)); \
                                     hipLaunchKernelGGL((dwconv7_lnb_kernel<CC, 2, 0, NWW, true>), grid, dim3(NWW * 64), ldsp, s, a, Sg, sprg, nstg); return 0; }
                        DWB_PACK(192, 6) DWB_PACK(192, 9) DWB_PACK(192, 12) DWB_PACK(384, 6) DWB_PACK(384, 9) DWB_PACK(384, 12)
#undef DWB_PACK
                    }
                }
                // 12 waves per block (three per SIMD) for C = 768 / 384: 177 -> 147 us and 313 -> 298 us on the stage-2 / stage-1 maps at 16 frames
                // (C = 192: 598 -> 676 us, stays on 8 waves).  UNI_DW_W12 = 0 / 1 / 2: off / default / also C = 192 (A/B switch).
                static const int w12 = getenv("UNI_DW_W12") ? atoi(getenv("UNI_DW_W12")) : 1;
                if (w12 && rows == 2 && (a.C == 384 || a.C == 768 || (w12 == 2 && a.C == 192))) {      // (C = 256 measured slower on 12 waves: 53.5 vs 45.2 us at 50 x 80)
                    const int Sw12 = 12 / wps;
                    const size_t lds12 = (size_t)49 * a.C * 4 + (size_t)2 * 12 * 16 * 4;
#define DWB_W12(CC) if (a.C == CC) { static DevOnce once; UNI_LDS_OPTIN(once, "dwconv7_lnb (12 waves)", 163840, reinterpret_cast<const void*>(&dwconv7_lnb_kernel<CC, 2, 0, 12>)); \
                                     hipLaunchKernelGGL((dwconv7_lnb_kernel<CC, 2, 0, 12>), grid, dim3(Sw12 * wps * 64), lds12, s, a, Sw12, spr, nst); return 0; }
                    DWB_W12(192) DWB_W12(384) DWB_W12(768)
#undef DWB_W12
                }
#define DWB_GO(CC, RR) if (a.C == CC && rows == RR) { hipLaunchKernelGGL((dwconv7_lnb_kernel<CC, RR>), grid, block, ldsw, s, a, Sw, spr, nst); return 0; }
                DWB_ALL(DWB_GO)
#undef DWB_GO
#undef DWB_ALL
            }
        }
        hipLaunchKernelGGL(dwconv7_ln2_kernel, dim3(cdiv(nstrips, S)), dim3(T), strip_reduce_lds(S, CG, 16), s, a, S, CG, spr, nstrips);
        return 0;
    }
    if (px == 8) {
        const int spr = cdiv(a.W, 8), nstrips = spr * a.H * nb;
        hipLaunchKernelGGL(dwconv7_ln_kernel<8>, dim3(cdiv(nstrips, S)), dim3(T), strip_reduce_lds(S, CG, 8), s, a, S, CG, spr, nstrips);
    } else {
        const int spr = cdiv(a.W, 4), nstrips = spr * a.H * nb;
        hipLaunchKernelGGL(dwconv7_ln_kernel<4>, dim3(cdiv(nstrips, S)), dim3(T), strip_reduce_lds(S, CG, 4), s, a, S, CG, spr, nstrips);
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// stem: conv 4x4 / stride 4 (3 -> C) + bias + LN over C.  thread = (pixel, 4 out channels)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void stem_kernel(StemArgs p, int S, int CG, int npix) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    const int cg = tid % CG, sl = tid / CG;
    const int Wo = p.W >> 2;
    const int pix = blockIdx.x * S + sl;
    const bool active = sl < S && pix < npix;
    float* patch = lds;                               // [S][48]
    float* red = lds + S * 48;
    // cooperative patch load: S*48 values, k = c*16 + ky*4 + kx
    for (int u = tid; u < S * 48; u += blockDim.x) {
        int sp = u / 48, k = u - sp * 48;
        int px = blockIdx.x * S + sp;
        float v = 0.f;
        if (px < npix) {
            const int per = (p.H >> 2) * Wo;
            int sb = px / per, q = px - sb * per;
            int oy = q / Wo, ox = q - oy * Wo;
            int c = k >> 4, ky = (k >> 2) & 3, kx = k & 3;
            v = p.img[(((size_t)sb * 3 + c) * p.H + oy * 4 + ky) * p.W + ox * 4 + kx];
        }
        patch[u] = v;
    }
    __syncthreads();
    float4 acc = *reinterpret_cast<const float4*>(p.bias + cg * 4);
    if (active) {
#pragma unroll 8
        for (int k = 0; k < 48; ++k) {
            float4 w = *reinterpret_cast<const float4*>(p.w + (size_t)k * p.C + cg * 4);
            float v = patch[sl * 48 + k];
            acc.x = fmaf(w.x, v, acc.x);
            acc.y = fmaf(w.y, v, acc.y);
            acc.z = fmaf(w.z, v, acc.z);
            acc.w = fmaf(w.w, v, acc.w);
        }
    }
    float part[1] = {acc.x + acc.y + acc.z + acc.w}, tot[1];
    strip_reduce<1>(part, tot, red, S, CG, sl, cg, active);
    const float mean = tot[0] / p.C;
    float a = acc.x - mean, b = acc.y - mean, c = acc.z - mean, d = acc.w - mean;
    part[0] = a * a + b * b + c * c + d * d;
    strip_reduce<1>(part, tot, red, S, CG, sl, cg, active);
    if (!active) return;
    const float rstd = 1.f / sqrtf(tot[0] / p.C + 1e-6f);
    const float4 g = *reinterpret_cast<const float4*>(p.gamma + cg * 4);
    const float4 be = *reinterpret_cast<const float4*>(p.beta + cg * 4);
    float4 o = make_float4(a * rstd * g.x + be.x, b * rstd * g.y + be.y, c * rstd * g.z + be.z, d * rstd * g.w + be.w);
    *reinterpret_cast<float4*>(p.out + (size_t)pix * p.C + cg * 4) = o;
}

// 4 output pixels (consecutive in x) x 4 channels per thread: the 48 weight float4s are loaded once per 4 pixels and the
// 4 x 48 patch values come from LDS as one float4 per tap (broadcast over the channel groups): 12 instead of ~50 global
// loads per output float4, 4x fewer blocks / barriers.  Needs W % 16 == 0 (launch_stem falls back otherwise).
__global__ __launch_bounds__(512) void stem4_kernel(StemArgs p, int SG, int CG, int nstrips, int spr) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    const int cg = tid % CG, sg = tid / CG;
    const int Ho = p.H >> 2;
    const int strip = blockIdx.x * SG + sg;
    const bool active = sg < SG && strip < nstrips;
    float* patch = lds;                               // [SG][48 taps][4 pixels]
    float* red = lds + SG * 192;
    // cooperative patch load: per strip 12 (c, ky) image rows x 4 pixels, one float4 (kx = 0..3 of one pixel) per task
    for (int u = tid; u < SG * 48; u += blockDim.x) {
        const int s_ = u / 48, r = u - s_ * 48;       // r = (c*4 + ky)*4 + q
        const int st = blockIdx.x * SG + s_;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (st < nstrips) {
            const int row = st / spr, xs = (st - row * spr) * 4;          // row over B*Ho, first of the 4 output pixels
            const int sb = row / Ho, oy = row - sb * Ho;
            const int q = r & 3, cky = r >> 2, c = cky >> 2, ky = cky & 3;
            v = *reinterpret_cast<const float4*>(p.img + (((size_t)sb * 3 + c) * p.H + oy * 4 + ky) * p.W + (xs + q) * 4);
        }
        const int q = r & 3, kbase = (r >> 2) * 4;    // taps kbase .. kbase+3 (kx) of pixel q
        float* dst = patch + s_ * 192 + q;
        dst[(kbase + 0) * 4] = v.x; dst[(kbase + 1) * 4] = v.y; dst[(kbase + 2) * 4] = v.z; dst[(kbase + 3) * 4] = v.w;
    }
    __syncthreads();
    f32x4 acc[4];
    {
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + cg * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = b;
    }
    if (active) {
        const float* pt = patch + sg * 192;
#pragma unroll 8
        for (int k = 0; k < 48; ++k) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(p.w + (size_t)k * p.C + cg * 4);
            const f32x4 v4 = *reinterpret_cast<const f32x4*>(pt + k * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[q][e] = fmaf(w[e], v4[q], acc[q][e]);
        }
    }
    float part[4], tot[4], mean[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) part[q] = acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    strip_reduce<4>(part, tot, red, SG, CG, sg, cg, active);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        mean[q] = tot[q] / p.C;
        float a = acc[q][0] - mean[q], b = acc[q][1] - mean[q], c = acc[q][2] - mean[q], d = acc[q][3] - mean[q];
        part[q] = a * a + b * b + c * c + d * d;
    }
    strip_reduce<4>(part, tot, red, SG, CG, sg, cg, active);
    if (!active) return;
    const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + cg * 4);
    const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + cg * 4);
    const int row = strip / spr, xs = (strip - row * spr) * 4;
    const size_t pix0 = (size_t)row * (p.W >> 2) + xs;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float rstd = 1.f / sqrtf(tot[q] / p.C + 1e-6f);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (acc[q][e] - mean[q]) * rstd * g[e] + be[e];
        *reinterpret_cast<f32x4*>(p.out + (pix0 + q) * p.C + cg * 4) = o;
    }
}

int launch_stem(const StemArgs& a, hipStream_t s) {
    UNI_REQUIRE(a.C % 4 == 0 && a.H % 4 == 0 && a.W % 4 == 0, "stem: C=%d H=%d W=%d unsupported", a.C, a.H, a.W);
    const int CG = a.C / 4;
    int S = 256 / CG;
    if (S < 1) S = 1;
    const int npix = (a.H / 4) * (a.W / 4) * (a.B > 0 ? a.B : 1);
    static const char* env = getenv("UNI_STEM1");
    if (a.W % 16 == 0 && !env) {
        const int spr = a.W / 16, nstrips = npix / 4;
        const int T4 = cdiv(S * CG, 64) * 64;
        size_t lds4 = S * 192 * sizeof(float) + strip_reduce_lds(S, CG, 4);
        hipLaunchKernelGGL(stem4_kernel, dim3(cdiv(nstrips, S)), dim3(T4), lds4, s, a, S, CG, nstrips, spr);
        return 0;
    }
    const int T = cdiv(S * CG, 64) * 64;
    size_t lds = S * 48 * sizeof(float) + strip_reduce_lds(S, CG, 1);
    hipLaunchKernelGGL(stem_kernel, dim3(cdiv(npix, S)), dim3(T), lds, s, a, S, CG, npix);
    return 0;
}
