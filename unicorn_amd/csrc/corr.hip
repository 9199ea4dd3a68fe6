// K4: dense embedding correlation + softmax over the REFERENCE axis + prior propagation, fused
// flash-style (external/lib/test/tracker/unicorn_sot.py:88-100, unicorn_vos.py:166-186):
//     out[k][q] = sum_r V[k][r] * softmax_r( <Eref[r,:], Ecur[q,:]> )          (no 1/sqrt(d) scaling)
// The HW x HW similarity (16000^2 at 800x1280, 1 GB in fp32) is never materialised: each wave keeps 32
// current-frame pixels (the MFMA "B" operand, 64 VGPRs) stationary, streams 32-row reference tiles through
// LDS, and because the MFMA accumulator gives every lane 16 reference rows of ONE query column, the
// online-softmax reduction over the reference axis is lane-local (one cross-half shuffle at the end).
// precision 0: exact fp32 on v_mfma_f32_32x32x2_f32 (157 TF peak, bitwise an fmaf chain).
// The reference axis is additionally split across blocks (flash-decoding style) to fill 256 CUs; a tiny
// second kernel merges the (max, sum, acc) partials.
#include "kernels.h"
#include <cstdlib>

namespace {
constexpr int CD = 128;      // embedding dim (unicorn.py:41-44)
constexpr int LDA = 132;     // padded LDS row stride (floats): 16 distinct rows -> 16 distinct 16-B bank slots
constexpr int TR = 32;       // reference rows per tile
constexpr int QB = 128;      // query columns per block (4 waves x 32)

__device__ __forceinline__ int rowof(int r, int fh) { return (r & 3) + 8 * (r >> 2) + 4 * fh; }

template <int KV>
__global__ __launch_bounds__(256, 2) void corr_f32_kernel(const float* __restrict__ eref,
                                                          const float* __restrict__ ecur,
                                                          const float* __restrict__ v, float* __restrict__ out,
                                                          float* __restrict__ ws, int R, int Q, int K, int nsplit,
                                                          int rows_per_split) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                    // [2][TR*LDA]
    float* Vs = smem + 2 * TR * LDA;     // [2][KV*TR]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int split = blockIdx.y;
    const int q = blockIdx.x * QB + wave * 32 + fr;
    const int qc = q < Q ? q : Q - 1;

    // stationary operand: this lane's half (fh) of the 128-d embedding of its query pixel.
    // K-permutation: MFMA step t contracts dims {t, 64+t} (lanes <32 carry t, lanes >=32 carry 64+t).
    float b[64];
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(ecur + (size_t)qc * CD + fh * 64);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            f32x4 t = src[i];
            b[4 * i] = t[0]; b[4 * i + 1] = t[1]; b[4 * i + 2] = t[2]; b[4 * i + 3] = t[3];
        }
    }
    const int r_begin = split * rows_per_split;
    const int r_end = min(R, r_begin + rows_per_split);
    const int ntiles = (r_end - r_begin + TR - 1) / TR;

    f32x4 ga[4];
    float gv;
    auto gload = [&](int t) {
        const int r0 = r_begin + t * TR;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int idx = tid + 256 * j;
            int row = min(r0 + (idx >> 5), R - 1);
            ga[j] = *reinterpret_cast<const f32x4*>(eref + (size_t)row * CD + (idx & 31) * 4);
        }
        const int k = tid >> 5, r = r0 + (tid & 31);
        gv = (k < K && k < KV && r < R) ? v[(size_t)k * R + r] : 0.f;
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int idx = tid + 256 * j;
            *reinterpret_cast<f32x4*>(As + buf * TR * LDA + (idx >> 5) * LDA + (idx & 31) * 4) = ga[j];
        }
        if (tid < KV * TR) Vs[buf * KV * TR + tid] = gv;
    };

    float m = -INFINITY, l = 0.f, o[KV];
#pragma unroll
    for (int k = 0; k < KV; ++k) o[k] = 0.f;

    gload(0);
    sstore(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) gload(t + 1);
        const float* arow = As + buf * TR * LDA + fr * LDA + fh * 64;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            f32x4 a4 = *reinterpret_cast<const f32x4*>(arow + 4 * i);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0], b[4 * i], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[1], b[4 * i + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[2], b[4 * i + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[3], b[4 * i + 3], acc, 0, 0, 0);
        }
        // online softmax over this tile's 16 rows owned by the lane
        const int r0 = r_begin + t * TR;
        float sc[16], tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sc[r] = (r0 + rowof(r, fh) < r_end) ? acc[r] : -INFINITY;
            tmax = fmaxf(tmax, sc[r]);
        }
        const float mn = fmaxf(m, tmax);
        if (mn > -INFINITY) {   // a lane whose 16 rows are all masked (ragged tail) keeps its state
            const float f = (m > -INFINITY) ? __expf(m - mn) : 0.f;
            l *= f;
#pragma unroll
            for (int k = 0; k < KV; ++k) o[k] *= f;
            const float* vt = Vs + buf * KV * TR;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float pr[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    pr[j] = __expf(sc[4 * g + j] - mn);
                    l += pr[j];
                }
#pragma unroll
                for (int k = 0; k < KV; ++k) {
                    f32x4 v4 = *reinterpret_cast<const f32x4*>(vt + k * TR + 8 * g + 4 * fh);
                    o[k] += pr[0] * v4[0] + pr[1] * v4[1] + pr[2] * v4[2] + pr[3] * v4[3];
                }
            }
            m = mn;
        }
        if (t + 1 < ntiles) sstore(buf ^ 1);
        __syncthreads();
    }
    // merge the two lane halves (rows 4*fh.. interleaved) of each query column
    {
        const float m2 = __shfl_xor(m, 32, 64), l2 = __shfl_xor(l, 32, 64);
        const float M = fmaxf(m, m2);
        const float f1 = (m > -INFINITY) ? __expf(m - M) : 0.f;
        const float f2 = (m2 > -INFINITY) ? __expf(m2 - M) : 0.f;
        l = l * f1 + l2 * f2;
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            const float o2 = __shfl_xor(o[k], 32, 64);
            o[k] = o[k] * f1 + o2 * f2;
        }
        m = M;
    }
    if (fh == 0 && q < Q) {
        if (nsplit == 1) {
#pragma unroll
            for (int k = 0; k < KV; ++k)
                if (k < K) out[(size_t)k * Q + q] = o[k] / l;
        } else {
            float* w = ws + ((size_t)split * Q + q) * (2 + KV);
            w[0] = m;
            w[1] = l;
#pragma unroll
            for (int k = 0; k < KV; ++k) w[2 + k] = o[k];
        }
    }
}

// ---- precision 1: fp32-equivalent contraction on the bf16 MFMA pipe ("bf16x3"): every fp32 operand is split exactly into
// three bf16 pieces x = h + m + l (8+8+8 significand bits); the six products h.h, h.m, m.h, h.l, m.m, l.h are each exact in
// fp32 and are accumulated in the fp32 MFMA accumulator; the dropped terms (m.l, l.m, l.l) are <= 2^-24 relative, i.e. at
// the level of fp32 rounding itself.  6 x v_mfma_f32_32x32x16_bf16 (32 cyc each) per 16-deep slice instead of
// 8 x v_mfma_f32_32x32x2_f32 (64 cyc each): 2.67x less MFMA time at fp32-class accuracy (the reference itself runs this
// product in fp16).  8 waves x 32 queries share one 32-row reference tile; the tile is split while it is staged:
// LDS = 3 planes [32 rows][128 dims] bf16, 16-B chunks XOR-swizzled with (row & 15) (conflict-free ds_read_b128).
constexpr int QB2 = 256;     // query columns per block in the 8-wave variant (NWV x 32)
__device__ __forceinline__ void split3(const float (&x)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const bf16 hh = (bf16)x[e];
        const float r1 = x[e] - (float)hh;
        const bf16 mm = (bf16)r1;
        const float r2 = r1 - (float)mm;
        h[e] = hh; m[e] = mm; l[e] = (bf16)r2;
    }
}

template <int KV, int NWV>
__global__ __launch_bounds__(64 * NWV) void corr_split_kernel(const float* __restrict__ eref, const float* __restrict__ ecur,
                                                         const float* __restrict__ v, float* __restrict__ out,
                                                         float* __restrict__ ws, int R, int Q, int K, int nsplit,
                                                         int rows_per_split) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int PLANE = TR * CD;                       // bf16 elements per piece plane (32 x 128)
    bf16* As = reinterpret_cast<bf16*>(smem);            // [2 buffers][3 planes][TR][CD]
    float* Vs = smem + (2 * 3 * PLANE) / 2;              // [2][KV*TR]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int split = blockIdx.y;
    const int q = blockIdx.x * (32 * NWV) + wave * 32 + fr;
    const int qc = q < Q ? q : Q - 1;

    // stationary operand: MFMA slice c contracts dims 16c .. 16c+15, this lane carries dims 16c + 8 fh + (0..7)
    bf16x8 qh[8], qm[8], ql[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const f32x4* src = reinterpret_cast<const f32x4*>(ecur + (size_t)qc * CD + 16 * c + 8 * fh);
        const f32x4 t0 = src[0], t1 = src[1];
        constexpr float L2E = 1.4426950408889634f;         // scores come out in the log2 domain (exp2 needs no pre-multiply)
        const float x[8] = {t0[0] * L2E, t0[1] * L2E, t0[2] * L2E, t0[3] * L2E, t1[0] * L2E, t1[1] * L2E, t1[2] * L2E, t1[3] * L2E};
        split3(x, qh[c], qm[c], ql[c]);
    }
    const int r_begin = split * rows_per_split;
    const int r_end = min(R, r_begin + rows_per_split);
    const int ntiles = (r_end - r_begin + TR - 1) / TR;

    // staging: the 32 x 128 tile is 512 chunks of 8 dims; thread -> chunks tid + 64*NWV*j (row = chunk/16, slot = chunk%16)
    constexpr int SPT = 512 / (64 * NWV);            // chunks per thread (1 or 2)
    f32x4 g0[SPT], g1[SPT];
    float gv;
    auto gload = [&](int t) __attribute__((always_inline)) {
        const int r0 = r_begin + t * TR;
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const int ch = tid + 64 * NWV * j;
            const int row = min(r0 + (ch >> 4), R - 1);
            const f32x4* src = reinterpret_cast<const f32x4*>(eref + (size_t)row * CD + (ch & 15) * 8);
            g0[j] = src[0]; g1[j] = src[1];
        }
        const int k = tid >> 5, r = r0 + (tid & 31);
        gv = (k < K && k < KV && r < R) ? v[(size_t)k * R + r] : 0.f;
    };
    auto sstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const int ch = tid + 64 * NWV * j, srow = ch >> 4, sch = ch & 15;
            const float x[8] = {g0[j][0], g0[j][1], g0[j][2], g0[j][3], g1[j][0], g1[j][1], g1[j][2], g1[j][3]};
            bf16x8 h, m, l;
            split3(x, h, m, l);
            bf16* dst = As + buf * 3 * PLANE + srow * CD + ((sch ^ (srow & 15)) << 3);
            *reinterpret_cast<bf16x8*>(dst) = h;
            *reinterpret_cast<bf16x8*>(dst + PLANE) = m;
            *reinterpret_cast<bf16x8*>(dst + 2 * PLANE) = l;
        }
        if (tid < KV * TR) Vs[buf * KV * TR + tid] = gv;
    };

    float m = -INFINITY, l = 0.f, o[KV];
#pragma unroll
    for (int k = 0; k < KV; ++k) o[k] = 0.f;

    gload(0);
    sstore(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) gload(t + 1);
        const bf16* arow = As + buf * 3 * PLANE + fr * CD;
        // two independent accumulator chains (small cross terms / leading terms) keep the MFMA pipe back-to-back
        f32x16 acc, acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int off = ((2 * c + fh) ^ (fr & 15)) << 3;
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(arow + off);
            const bf16x8 am = *reinterpret_cast<const bf16x8*>(arow + PLANE + off);
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(arow + 2 * PLANE + off);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, qh[c], acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qh[c], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, qm[c], acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qm[c], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ql[c], acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, qh[c], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
        // online softmax over this tile's 16 rows owned by the lane, log2 domain; rows past the split end exist only in
        // its last tile, so the mask selects are skipped on full tiles (wave-uniform branch)
        const int r0 = r_begin + t * TR;
        float sc[16], tmax = -INFINITY;
        if (r0 + TR <= r_end) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { sc[r] = acc[r]; tmax = fmaxf(tmax, sc[r]); }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sc[r] = (r0 + rowof(r, fh) < r_end) ? acc[r] : -INFINITY;
                tmax = fmaxf(tmax, sc[r]);
            }
        }
        const float mn = fmaxf(m, tmax);
        if (mn > -INFINITY) {
            const float f = (m > -INFINITY) ? __builtin_amdgcn_exp2f(m - mn) : 0.f;
            l *= f;
#pragma unroll
            for (int k = 0; k < KV; ++k) o[k] *= f;
            const float* vt = Vs + buf * KV * TR;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float pr[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    pr[j] = __builtin_amdgcn_exp2f(sc[4 * g + j] - mn);
                    l += pr[j];
                }
#pragma unroll
                for (int k = 0; k < KV; ++k) {
                    f32x4 v4 = *reinterpret_cast<const f32x4*>(vt + k * TR + 8 * g + 4 * fh);
                    o[k] += pr[0] * v4[0] + pr[1] * v4[1] + pr[2] * v4[2] + pr[3] * v4[3];
                }
            }
            m = mn;
        }
        if (t + 1 < ntiles) sstore(buf ^ 1);
        __syncthreads();
    }
    m *= 0.6931471805599453f;                              // back to the natural-log domain of the merge (-inf stays -inf)
    {
        const float m2 = __shfl_xor(m, 32, 64), l2 = __shfl_xor(l, 32, 64);
        const float M = fmaxf(m, m2);
        const float f1 = (m > -INFINITY) ? __expf(m - M) : 0.f;
        const float f2 = (m2 > -INFINITY) ? __expf(m2 - M) : 0.f;
        l = l * f1 + l2 * f2;
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            const float o2 = __shfl_xor(o[k], 32, 64);
            o[k] = o[k] * f1 + o2 * f2;
        }
        m = M;
    }
    if (fh == 0 && q < Q) {
        if (nsplit == 1) {
#pragma unroll
            for (int k = 0; k < KV; ++k)
                if (k < K) out[(size_t)k * Q + q] = o[k] / l;
        } else {
            float* w = ws + ((size_t)split * Q + q) * (2 + KV);
            w[0] = m;
            w[1] = l;
#pragma unroll
            for (int k = 0; k < KV; ++k) w[2 + k] = o[k];
        }
    }
}

// ---- precision 2: fp32-equivalent contraction on the f16 MFMA pipe ("f16x2", the operand format of the f16x2 model precision):
// x = h + l with h = f16(x), l = f16(x - h) (22 significand bits; embeddings are O(1..100), far inside the f16 range; values below
// 2^-14 keep an ABSOLUTE error of 2^-25).  Three products h.h, h.l, l.h per 16-deep slice (v_mfma_f32_32x32x16_f16, fp32
// accumulate) instead of the six of the bf16x3 split: half the MFMA time at the same error class (the dropped l.l term is
// <= 2^-22 relative).  Same structure as corr_split_kernel; LDS = 2 planes [32 rows][128 dims] f16 per buffer.
__device__ __forceinline__ void split2h(const float (&x)[8], f16x8& h, f16x8& l) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        f16 hh, ll;
        h2_split(x[e], hh, ll);
        h[e] = hh; l[e] = ll;
    }
}

// H1 = the reference driver's own arithmetic class (unicorn_sot.py:95-100 casts keys, queries and values to fp16): operands rounded to
// f16 (the hi halves only), ONE MFMA per product, scores rounded to f16 before the softmax.  (The driver's `trans.half()` rounding of
// the NORMALISED softmax needs the final row sums, i.e. a second pass, and is not reproduced: probabilities stay fp32.)
template <int KV, int NWV, bool H1 = false>
__global__ __launch_bounds__(64 * NWV) void corr_h2_kernel(const float* __restrict__ eref, const float* __restrict__ ecur,
                                                      const float* __restrict__ v, float* __restrict__ out,
                                                      float* __restrict__ ws, int R, int Q, int K, int nsplit,
                                                      int rows_per_split, long v_frame_stride) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    {   // blockIdx.z = frame of a batch: [B][R][128] / [B][Q][128] embeddings, [B][K][Q] outputs, value rows shared (stride 0) or per frame
        const size_t z = blockIdx.z;
        eref += z * (size_t)R * CD;
        ecur += z * (size_t)Q * CD;
        out += z * (size_t)K * Q;
        ws += z * (size_t)nsplit * Q * (2 + KV);
        v += z * (size_t)v_frame_stride;
    }
    constexpr int PLANE = TR * CD;                       // f16 elements per piece plane (32 x 128)
    f16* As = reinterpret_cast<f16*>(smem);              // [2 buffers][2 planes][TR][CD]
    float* Vs = smem + (2 * 2 * PLANE) / 2;              // [2][KV*TR]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int split = blockIdx.y;
    const int q = blockIdx.x * (32 * NWV) + wave * 32 + fr;
    const int qc = q < Q ? q : Q - 1;

    f16x8 qh[8], ql[8];                                  // stationary operand: slice c = dims 16c + 8 fh + (0..7), log2 domain
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const f32x4* src = reinterpret_cast<const f32x4*>(ecur + (size_t)qc * CD + 16 * c + 8 * fh);
        const f32x4 t0 = src[0], t1 = src[1];
        constexpr float L2E = H1 ? 1.f : 1.4426950408889634f;    // H1: natural domain, the f16-rounded score is scaled afterwards
        const float x[8] = {t0[0] * L2E, t0[1] * L2E, t0[2] * L2E, t0[3] * L2E, t1[0] * L2E, t1[1] * L2E, t1[2] * L2E, t1[3] * L2E};
        split2h(x, qh[c], ql[c]);
    }
    const int r_begin = split * rows_per_split;
    const int r_end = min(R, r_begin + rows_per_split);
    const int ntiles = (r_end - r_begin + TR - 1) / TR;

    constexpr int SPT = 512 / (64 * NWV);
    f32x4 g0[SPT], g1[SPT];
    float gv;
    auto gload = [&](int t) __attribute__((always_inline)) {
        const int r0 = r_begin + t * TR;
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const int ch = tid + 64 * NWV * j;
            const int row = min(r0 + (ch >> 4), R - 1);
            const f32x4* src = reinterpret_cast<const f32x4*>(eref + (size_t)row * CD + (ch & 15) * 8);
            g0[j] = src[0]; g1[j] = src[1];
        }
        const int k = tid >> 5, r = r0 + (tid & 31);
        gv = (k < K && k < KV && r < R) ? v[(size_t)k * R + r] : 0.f;
        if (H1) gv = (float)(f16)gv;
    };
    auto sstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const int ch = tid + 64 * NWV * j, srow = ch >> 4, sch = ch & 15;
            const float x[8] = {g0[j][0], g0[j][1], g0[j][2], g0[j][3], g1[j][0], g1[j][1], g1[j][2], g1[j][3]};
            f16x8 h, l;
            split2h(x, h, l);
            f16* dst = As + buf * 2 * PLANE + srow * CD + ((sch ^ (srow & 15)) << 3);
            *reinterpret_cast<f16x8*>(dst) = h;
            *reinterpret_cast<f16x8*>(dst + PLANE) = l;
        }
        if (tid < KV * TR) Vs[buf * KV * TR + tid] = gv;
    };

    float m = -INFINITY, l = 0.f, o[KV];
#pragma unroll
    for (int k = 0; k < KV; ++k) o[k] = 0.f;

    gload(0);
    sstore(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) gload(t + 1);
        const f16* arow = As + buf * 2 * PLANE + fr * CD;
        f32x16 acc, acc2;                                  // two independent chains (leading / cross terms)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int off = ((2 * c + fh) ^ (fr & 15)) << 3;
            const f16x8 ah = *reinterpret_cast<const f16x8*>(arow + off);
            const f16x8 al = *reinterpret_cast<const f16x8*>(arow + PLANE + off);
            if (!H1) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[c], acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[c], acc, 0, 0, 0);
            if (!H1) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[c], acc2, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = H1 ? (float)(f16)acc[r] * 1.4426950408889634f : acc[r] + acc2[r];
        const int r0 = r_begin + t * TR;
        float sc[16], tmax = -INFINITY;
        if (r0 + TR <= r_end) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { sc[r] = acc[r]; tmax = fmaxf(tmax, sc[r]); }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sc[r] = (r0 + rowof(r, fh) < r_end) ? acc[r] : -INFINITY;
                tmax = fmaxf(tmax, sc[r]);
            }
        }
        const float mn = fmaxf(m, tmax);
        if (mn > -INFINITY) {
            const float f = (m > -INFINITY) ? __builtin_amdgcn_exp2f(m - mn) : 0.f;
            l *= f;
#pragma unroll
            for (int k = 0; k < KV; ++k) o[k] *= f;
            const float* vt = Vs + buf * KV * TR;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float pr[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    pr[j] = __builtin_amdgcn_exp2f(sc[4 * g + j] - mn);
                    l += pr[j];
                }
#pragma unroll
                for (int k = 0; k < KV; ++k) {
                    f32x4 v4 = *reinterpret_cast<const f32x4*>(vt + k * TR + 8 * g + 4 * fh);
                    o[k] += pr[0] * v4[0] + pr[1] * v4[1] + pr[2] * v4[2] + pr[3] * v4[3];
                }
            }
            m = mn;
        }
        if (t + 1 < ntiles) sstore(buf ^ 1);
        __syncthreads();
    }
    m *= 0.6931471805599453f;
    {
        const float m2 = __shfl_xor(m, 32, 64), l2 = __shfl_xor(l, 32, 64);
        const float M = fmaxf(m, m2);
        const float f1 = (m > -INFINITY) ? __expf(m - M) : 0.f;
        const float f2 = (m2 > -INFINITY) ? __expf(m2 - M) : 0.f;
        l = l * f1 + l2 * f2;
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            const float o2 = __shfl_xor(o[k], 32, 64);
            o[k] = o[k] * f1 + o2 * f2;
        }
        m = M;
    }
    if (fh == 0 && q < Q) {
        if (nsplit == 1) {
#pragma unroll
            for (int k = 0; k < KV; ++k)
                if (k < K) out[(size_t)k * Q + q] = o[k] / l;
        } else {
            float* w = ws + ((size_t)split * Q + q) * (2 + KV);
            w[0] = m;
            w[1] = l;
#pragma unroll
            for (int k = 0; k < KV; ++k) w[2 + k] = o[k];
        }
    }
}

template <int KV>
__global__ void corr_merge_kernel(const float* __restrict__ ws, float* __restrict__ out, int Q, int K, int nsplit) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    ws += (size_t)blockIdx.y * nsplit * Q * (2 + KV);          // blockIdx.y = frame of a batch
    out += (size_t)blockIdx.y * K * Q;
    float M = -INFINITY;
    for (int s = 0; s < nsplit; ++s) M = fmaxf(M, ws[((size_t)s * Q + q) * (2 + KV)]);
    float L = 0.f, O[KV];
#pragma unroll
    for (int k = 0; k < KV; ++k) O[k] = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float* w = ws + ((size_t)s * Q + q) * (2 + KV);
        const float f = (w[0] > -INFINITY) ? __expf(w[0] - M) : 0.f;
        L += w[1] * f;
#pragma unroll
        for (int k = 0; k < KV; ++k) O[k] += w[2 + k] * f;
    }
#pragma unroll
    for (int k = 0; k < KV; ++k)
        if (k < K) out[(size_t)k * Q + q] = O[k] / L;
}

int corr_slots(int precision) {     // co-resident blocks on the device: 2 x 4-wave blocks (fp32) / 1 x 8-wave block (split) per CU
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                  ? prop.multiProcessorCount : 256;
    }
    return precision ? ncu : 2 * ncu;
}
// Split of the reference axis: every block walks rows/ns rows in 32-row tiles; blocks run in rounds of `slots`, so the
// cost model is rounds x (tiles per split + fixed per-block overhead).  (A fixed ">= 1024 blocks" target gave 4.2 rounds
// = 5 at 800x1280, 16 % quantisation loss.)
int pick_nsplit(int R, int Q, int precision = 0, int B = 1) {
    const int nqb = cdiv(Q, precision ? QB2 : QB) * B;
    static const char* env = getenv("UNI_CORR_BLOCKS");
    int maxs = R / 256;               // keep >= 8 tiles per split
    if (maxs < 1) maxs = 1;
    if (env) {
        int ns = cdiv(atoi(env), nqb);
        return ns > maxs ? maxs : (ns < 1 ? 1 : ns);
    }
    const int slots = corr_slots(precision);
    int best = 1;
    long best_cost = -1;
    for (int ns = 1; ns <= maxs && ns <= 64; ++ns) {
        const long rounds = cdiv(nqb * ns, slots);
        const long cost = rounds * (cdiv(cdiv(R, ns), TR) + 4);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = ns; }
    }
    return best;
}

template <int KV>
int run(const float* eref, const float* ecur, const float* v, float* out, int R, int Q, int K, float* ws, size_t ws_bytes, int precision,
        hipStream_t s, int B = 1, long vfs = 0) {
    const int ns = pick_nsplit(R, Q, precision, B);
    int rps = cdiv(cdiv(R, ns), TR) * TR;
    const int ns_eff = cdiv(R, rps);   // every split non-empty
    // the partial results of THIS launch (real precision, real split) must fit the scratch the caller sized with corr_workspace_bytes*:
    // checked here, next to the launch, so a later per-precision change of pick_nsplit / corr_slots cannot write past it
    UNI_REQUIRE(ns_eff == 1 || (size_t)B * ns_eff * Q * (2 + KV) * sizeof(float) <= ws_bytes,
                "corr: workspace too small for %d splits x %d frames x %d queries (%zu B given)", ns_eff, B, Q, ws_bytes);
    if (precision == 3) {
        size_t lds = (size_t)2 * 2 * TR * CD * sizeof(f16) + (size_t)2 * KV * TR * sizeof(float);
        hipLaunchKernelGGL((corr_h2_kernel<KV, 8, true>), dim3(cdiv(Q, QB2), ns_eff, B), dim3(512), lds, s, eref, ecur, v, out, ws, R,
                           Q, K, ns_eff, rps, vfs);
    } else if (precision == 2) {
        size_t lds = (size_t)2 * 2 * TR * CD * sizeof(f16) + (size_t)2 * KV * TR * sizeof(float);
        hipLaunchKernelGGL((corr_h2_kernel<KV, 8>), dim3(cdiv(Q, QB2), ns_eff, B), dim3(512), lds, s, eref, ecur, v, out, ws, R,
                           Q, K, ns_eff, rps, vfs);
    } else if constexpr (KV > 8) {
        uni_set_error("corr: 16 value rows per pass only in precision 2 / 3");
        return -1;
    } else if (precision) {
        size_t lds = (size_t)2 * 3 * TR * CD * sizeof(bf16) + (size_t)2 * KV * TR * sizeof(float);
        hipLaunchKernelGGL((corr_split_kernel<KV, 8>), dim3(cdiv(Q, QB2), ns_eff), dim3(512), lds, s, eref, ecur, v, out, ws, R,
                           Q, K, ns_eff, rps);
    } else {
        size_t lds = (size_t)(2 * TR * LDA + 2 * KV * TR) * sizeof(float);
        hipLaunchKernelGGL((corr_f32_kernel<KV>), dim3(cdiv(Q, QB), ns_eff), dim3(256), lds, s, eref, ecur, v, out, ws, R,
                           Q, K, ns_eff, rps);
    }
    if (ns_eff > 1)
        hipLaunchKernelGGL((corr_merge_kernel<KV>), dim3(cdiv(Q, 256), B), dim3(256), 0, s, ws, out, Q, K, ns_eff);
    return 0;
}
}  // namespace

size_t corr_workspace_bytes(int R, int Q, int K) {
    (void)K;
    int m = 1;
    for (int prec = 0; prec <= 3; ++prec) m = max(m, pick_nsplit(R, Q, prec));      // every precision the launcher accepts
    return (size_t)m * Q * (2 + 16) * sizeof(float);
}

size_t corr_workspace_bytes_batched(int B, int R, int Q, int K) {
    (void)K;
    size_t m = corr_workspace_bytes(R, Q, K);                                    // the per-frame fall-back reuses one frame's scratch
    int ns = 1;
    for (int prec = 2; prec <= 3; ++prec) ns = max(ns, pick_nsplit(R, Q, prec, B));   // the precisions that run batched (others: frame by frame)
    const size_t b = (size_t)ns * B * Q * (2 + 16) * sizeof(float);
    return b > m ? b : m;
}

// B frames in ONE launch (blockIdx.z = frame): embeddings [B][R][128] / [B][Q][128], out [B][K][Q], value rows [K][R] shared by the frames
// (values_per_frame = 0: the SOT label map of the cached first frame) or [B][K][R].  The split of the reference axis is chosen for
// all B x Q / 256 blocks together: at 16 frames of 800 x 1280 one block walks the whole reference map (no partial results, no merge pass)
// -- 0.222 vs 0.236-0.243 ms per frame against 16 launches (tools/corr_batch_probe.py).  Precisions 0 / 1 and more than 16 value rows
// run frame by frame.
int launch_corr_batched(const float* eref, const float* ecur, const float* v, float* out, int B, int R, int Q, int D, int K,
                        int values_per_frame, int precision, void* workspace, size_t ws_bytes, hipStream_t s) {
    UNI_REQUIRE(B > 0, "corr: empty batch");
    UNI_REQUIRE(ws_bytes >= corr_workspace_bytes_batched(B, R, Q, K), "corr: workspace too small");
    const long vfs = values_per_frame ? (long)K * R : 0;
    if (B == 1 || precision < 2 || K > 16) {
        for (int b = 0; b < B; ++b) {
            const int rc = launch_corr(eref + (size_t)b * R * D, ecur + (size_t)b * Q * D, v + (size_t)b * vfs, out + (size_t)b * K * Q, R, Q, D, K,
                                       precision, workspace, ws_bytes, s);
            if (rc) return rc;
        }
        return 0;
    }
    UNI_REQUIRE(D == CD, "corr: embedding dim %d unsupported (128)", D);
    UNI_REQUIRE(R > 0 && Q > 0 && K > 0, "corr: empty problem R=%d Q=%d K=%d", R, Q, K);
    UNI_REQUIRE(precision <= 3, "corr: precision %d not implemented", precision);
    UNI_REQUIRE(((uintptr_t)eref & 15) == 0 && ((uintptr_t)ecur & 15) == 0, "corr: embeddings must be 16-B aligned");
    float* ws = reinterpret_cast<float*>(workspace);
    if (K == 1) return run<1>(eref, ecur, v, out, R, Q, K, ws, ws_bytes, precision, s, B, vfs);
    if (K <= 4) return run<4>(eref, ecur, v, out, R, Q, K, ws, ws_bytes, precision, s, B, vfs);
    if (K <= 8) return run<8>(eref, ecur, v, out, R, Q, K, ws, ws_bytes, precision, s, B, vfs);
    return run<16>(eref, ecur, v, out, R, Q, K, ws, ws_bytes, precision, s, B, vfs);
}

int launch_corr(const float* eref, const float* ecur, const float* v, float* out, int R, int Q, int D, int K,
                int precision, void* workspace, size_t ws_bytes, hipStream_t s) {
    UNI_REQUIRE(D == CD, "corr: embedding dim %d unsupported (128)", D);
    UNI_REQUIRE(R > 0 && Q > 0 && K > 0, "corr: empty problem R=%d Q=%d K=%d", R, Q, K);
    UNI_REQUIRE(precision >= 0 && precision <= 3, "corr: precision %d not implemented (0 = fp32 MFMA, 1 = bf16x3 split, 2 = f16x2 split, 3 = fp16 single pass)", precision);
    UNI_REQUIRE(ws_bytes >= corr_workspace_bytes(R, Q, K), "corr: workspace too small");
    UNI_REQUIRE(((uintptr_t)eref & 15) == 0 && ((uintptr_t)ecur & 15) == 0, "corr: embeddings must be 16-B aligned");
    float* ws = reinterpret_cast<float*>(workspace);
    // value rows (objects) in chunks of up to 16: every chunk re-evaluates the R x Q scores, so a VOS group of 9..16 objects
    // (unicorn_vos.py:166-186) costs ONE pass instead of two
    const int chunk = (precision >= 2) ? 16 : 8;      // the 16-row instantiation of the bf16x3 / exact-fp32 kernels spills: they keep 8
    for (int k0 = 0; k0 < K; k0 += chunk) {
        const int kc = K - k0 < chunk ? K - k0 : chunk;
        int rc;
        if (kc == 1) rc = run<1>(eref, ecur, v + (size_t)k0 * R, out + (size_t)k0 * Q, R, Q, kc, ws, ws_bytes, precision, s);
        else if (kc <= 4) rc = run<4>(eref, ecur, v + (size_t)k0 * R, out + (size_t)k0 * Q, R, Q, kc, ws, ws_bytes, precision, s);
        else if (kc <= 8) rc = run<8>(eref, ecur, v + (size_t)k0 * R, out + (size_t)k0 * Q, R, Q, kc, ws, ws_bytes, precision, s);
        else rc = run<16>(eref, ecur, v + (size_t)k0 * R, out + (size_t)k0 * Q, R, Q, kc, ws, ws_bytes, precision, s);
        if (rc) return rc;
    }
    return 0;
}
