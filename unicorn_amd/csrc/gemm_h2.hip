// K1b / K2 / K5 in the "f16x2" precision mode: fp32-EQUIVALENT implicit GEMM on the f16 MFMA pipe.
//
//   out[m][n] = epilogue( wscale * sum_k A[m][k] * W[n][k] )
//
// Both operands are stored split: x = hi + lo with hi = f16(x), lo = f16(x - hi) (22 significand bits; FMT_H2 in
// common.h: per 8 consecutive k a 32-byte group [8 x hi][8 x lo]).  Every product is evaluated as
//     hi.hi + hi.lo + lo.hi           (3 x v_mfma_f32_32x32x16_f16, each partial product exact in fp32, fp32 accumulate)
// and the dropped lo.lo term is <= 2^-22 relative: the result is in the error class of an fp32 fmaf chain at 3/16 of
// the MFMA time of the exact v_mfma_f32_32x32x2_f32 path (16 x the bf16 cost).  Why this format and not bf16 or bf16x2:
// oracle/error_budget.py / profiles/r02_precision_budget.json (box IoU >= 0.999 needs >= ~20 operand bits on EVERY
// stage with the synthetic weights; bf16x2 = 16 bits misses it, f16x2 passes on tiny and large).
// Weights are pre-scaled by a power of two per packed tensor (GemmArgs::wscale undoes it) so their lo halves stay in
// the normal f16 range; activations need no scaling (hi is clamped to +-65504, lo extends the range to 2 x that).
//
// Kernel structure = gemm.hip's: LDS-DMA double buffer (global_load_lds_dwordx4, source-side XOR swizzle), one barrier
// per K step, swapped MFMA operands, shared epilogue (gemm_epi.h).  A K step covers 32 k (128-byte LDS rows, the same
// geometry as the 64-k bf16 rows): chunk 2c of a row holds the hi halves of k = 8c..8c+7, chunk 2c+1 the lo halves.
// Per step and wave tile (64 x 64): 16 ds_read_b128 feed 24 MFMAs (bf16: 16 feed 16), i.e. the split kernel is closer to
// MFMA-bound than the bf16 one.
#include "kernels.h"
#include <cmath>
#include <cstdlib>
#include <cstring>

__device__ u32x4 g_zero_page_h2 = {0u, 0u, 0u, 0u};

#define GLDS16H(gptr, lptr)                                                                            \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),            \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)
#define OPAQUE64H(x)                                     \
    do {                                                 \
        int _lo = (int)(x), _hi = (int)((x) >> 32);      \
        asm volatile("" : "+v"(_lo), "+v"(_hi));         \
        (x) = ((long)_hi << 32) | (unsigned)_lo;         \
    } while (0)

#include "gemm_epi.h"

template <int WM, int WN, int TM, int TN, bool CONV, bool STATS, int BKE = 32>
__global__ __launch_bounds__(64 * WM * WN, (BKE == 16 ? 2 : 1)) void gemm_h2_kernel(GemmArgs p) {
    constexpr int NW = WM * WN;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int ROWB = 4 * BKE;            // LDS row bytes = BKE (k per step: 32 or 16) x (2 B hi + 2 B lo)
    constexpr int CPR = ROWB / 16;           // 16-B chunks per LDS row (8 or 4)
    constexpr int RPP = 64 / CPR;            // rows per 1-KiB DMA piece (8 or 16)
    constexpr int SW = BKE == 32 ? 1 : 2;    // swizzle: chunk' = chunk ^ ((row >> SW) & (CPR-1))  (256-B bank period, see gemm.hip)
    constexpr int A_PC = BM / RPP / NW, B_PC = BN / RPP / NW;
    static_assert(A_PC >= 1 && B_PC >= 1, "tile too small for the wave grid");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                                   // [2][BM][128 B]
    char* Bs = smem + 2 * BM * ROWB;                   // [2][BN][128 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int nbn = (p.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int L;
    {   // XCD-aware bijective block remap (block b runs on XCD b % 8)
        const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    int bm, bn;
    {   // super-tile order: chunks of 8 N tiles, M-major inside a chunk (see gemm.hip)
        const int nbm = (p.M + BM - 1) / BM;
        constexpr int GN = 8;
        const int per_chunk = nbm * GN;
        const int c = L / per_chunk;
        const int wc = min(GN, nbn - c * GN);
        const int rem = L - c * per_chunk;
        bm = rem / wc;
        bn = c * GN + rem - bm * wc;
    }
    const int m0 = bm * BM, n0 = bn * BN;

    // ---- per-lane DMA sources: lane -> row lane/8 of its piece, physical chunk lane%8, logical chunk = phys ^ ((row>>1)&7);
    // logical chunk lc -> k group lc>>1 (8 k), half lc&1 (0 = hi, 1 = lo)
    const int lrow = lane / CPR;
    const int lch = (lane % CPR) ^ (((RPP * wave + lrow) >> SW) & (CPR - 1));
    const int kgrp = lch >> 1, half = lch & 1;
    const char* abase = reinterpret_cast<const char*>(p.A);
    const long zoff = reinterpret_cast<const char*>(&g_zero_page_h2) - abase;
    int a_pix[A_PC];
#pragma unroll
    for (int i = 0; i < A_PC; ++i) {
        int m = m0 + RPP * (wave + NW * i) + lrow;
        m = m < p.M ? m : p.M - 1;
        if (CONV) {
            int b = m / p.Mper, q = m - b * p.Mper;
            int oy = q / p.Wout, ox = q - oy * p.Wout;
            a_pix[i] = (b << 24) | (oy << 12) | ox;
        } else {
            a_pix[i] = m;
        }
    }
    const char* wbase = reinterpret_cast<const char*>(p.W) + ((size_t)(n0 + RPP * wave + lrow) * p.Kpad) * 4 + lch * 16;

    auto issue = [&](int kt, int buf) {
        const int k = kt * BKE + kgrp * 8;                 // first of this lane's 8 k
        const bool kok = k < p.K;
        char* adst = As + buf * BM * ROWB + wave * 1024;
        char* bdst = Bs + buf * BN * ROWB + wave * 1024;
        if (CONV) {
            int tap = k / p.Cin;
            int c = k - tap * p.Cin;
            int ky = tap / p.KW, kx = tap - ky * p.KW;
#pragma unroll
            for (int i = 0; i < A_PC; ++i) {
                int bb = a_pix[i] >> 24, oy = (a_pix[i] >> 12) & 0xfff, ox = a_pix[i] & 0xfff;
                int iy = oy * p.stride - p.pad + ky, ix = ox * p.stride - p.pad + kx;
                bool ok = kok && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
                long off = ok ? (long)(((size_t)((bb * p.Hin + iy) * p.Win + ix) * p.lda + c) * 4 + half * 16) : zoff;
                OPAQUE64H(off);
                GLDS16H(abase + off, adst + i * NW * 1024);
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_PC; ++i) {
                long off = kok ? (long)(((size_t)a_pix[i] * p.lda + k) * 4 + half * 16) : zoff;
                OPAQUE64H(off);
                GLDS16H(abase + off, adst + i * NW * 1024);
            }
        }
#pragma unroll
        for (int i = 0; i < B_PC; ++i) GLDS16H(wbase + ((size_t)i * NW * RPP * p.Kpad + (size_t)kt * BKE) * 4, bdst + i * NW * 1024);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int kt0 = 0, nk = (p.K + BKE - 1) / BKE;           // the packed weights are zero beyond K (Kpad >= roundup(K, 64))
    if (p.splitk > 1) {                                // split-K: this block owns K steps [kt0, nk) of its tile
        const int per = (nk + p.splitk - 1) / p.splitk;
        kt0 = blockIdx.y * per;
        nk = min(nk, kt0 + per);       // an empty range (kt0 >= nk) still stores its (zero) partial tile: the reduce kernel reads every slab
    }
    const int fr = lane & 31, fh = lane >> 5;
    if (kt0 < nk) issue(kt0, 0);
    for (int kt = kt0; kt < nk; ++kt) {
        const int buf = (kt - kt0) & 1;
        __syncthreads();                                   // K slice kt landed (vmcnt(0) + barrier); everyone is done with buf^1
        if (kt + 1 < nk && !(p.dbg & 1)) issue(kt + 1, buf ^ 1);     // dbg 1: ablation, no DMA after slice 0
        if (p.dbg & 2) continue;                                     // dbg 2: ablation, DMA only
        const char* a = As + buf * BM * ROWB;
        const char* b = Bs + buf * BN * ROWB;
        // 16 k per MFMA; lane half fh carries k = 16 kk + 8 fh + (0..7); logical chunk 2 (2 kk + fh) holds the hi halves, +1 the lo
        f16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
        auto ldfrag = [&](int kk, int slot) __attribute__((always_inline)) {
            const int ch = 2 * (2 * kk + fh);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * 32 * TM + i * 32 + fr, sw = (row >> SW) & (CPR - 1);
                ah[slot][i] = *reinterpret_cast<const f16x8*>(a + row * ROWB + ((ch ^ sw) << 4));
                al[slot][i] = *reinterpret_cast<const f16x8*>(a + row * ROWB + (((ch + 1) ^ sw) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn * 32 * TN + j * 32 + fr, sw = (row >> SW) & (CPR - 1);
                bh[slot][j] = *reinterpret_cast<const f16x8*>(b + row * ROWB + ((ch ^ sw) << 4));
                bl[slot][j] = *reinterpret_cast<const f16x8*>(b + row * ROWB + (((ch + 1) ^ sw) << 4));
            }
        };
        // PF (8-wave 256 x 256 tile, 2 waves per SIMD): the fragments of sub-step 1 are read while the 24 MFMAs of sub-step 0
        // issue (register double buffer); the 4-waves-per-SIMD tiles rely on wave interleaving instead (128-VGPR budget)
        constexpr bool PF = TM * TN >= 8 && BKE == 32;
        ldfrag(0, 0);
#pragma unroll
        for (int kk = 0; kk < BKE / 16; ++kk) {
            const int sl = PF ? kk : 0;
            if (kk == 0 && PF) ldfrag(1, 1);
            if (kk == 1 && !PF) ldfrag(1, 0);
            // swapped operands (weights = MFMA A): lane -> pixel row, 4 consecutive channels per accumulator quad
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[sl][j], ah[sl][i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[sl][j], al[sl][i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[sl][j], ah[sl][i], acc[i][j], 0, 0, 0);
        }
    }
    if (p.dbg & 16) {   // ablation: no epilogue (accumulators kept live)
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
        if (sacc == 1.2345e-30f && p.outF) p.outF[0] = sacc;
        return;
    }
    if (p.wscale != 1.f) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= p.wscale;
    }
    if (!STATS && p.splitk > 1) {   // partial product of one K range -> its slab; bias / residual / statistics belong to the reduce kernel
        GemmArgs q = p;
        q.outF = p.slab + (size_t)blockIdx.y * p.M * p.N; q.ldf = p.N;
        q.bias = nullptr; q.res = nullptr; q.outB = nullptr; q.stats = nullptr; q.out_hw = 0; q.act = ACT_NONE;
        gemm_epilogue<WM, WN, TM, TN, false>(q, acc, m0, n0, wm, wn, lane, tid, smem);
        return;
    }
    gemm_epilogue<WM, WN, TM, TN, STATS>(p, acc, m0, n0, wm, wn, lane, tid, smem);
}

// split-K reduce: out[m][n] = sum_s slab[s][m][n] + bias[n] (+ res[m][n]); optional GroupNorm group sums of the result per sample.
// Block = RB rows of ONE sample x all N columns (thread -> float4 column groups); the RB x splitk loads of a thread are independent
// (the first version walked 32 rows x splits serially per thread: latency-bound, slower than the GEMM it served).
constexpr int RED_RB = 8;
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs p) {
    __shared__ float gacc[64];
    const int sb = blockIdx.y, r0 = blockIdx.x * RED_RB;
    if (p.stats) {
        if (threadIdx.x < 64) gacc[threadIdx.x] = 0.f;
        __syncthreads();
    }
    const size_t MN = (size_t)p.M * p.N;
    for (int c = threadIdx.x * 4; c < p.N; c += 1024) {
        f32x4 v[RED_RB];
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + c);
#pragma unroll
        for (int i = 0; i < RED_RB; ++i) v[i] = bv;
        for (int k = 0; k < p.splitk; ++k) {
            const float* sk = p.slab + k * MN + ((size_t)sb * p.Mper + r0) * p.N + c;
#pragma unroll
            for (int i = 0; i < RED_RB; ++i)
                if (r0 + i < p.Mper) v[i] += *reinterpret_cast<const f32x4*>(sk + (size_t)i * p.N);
        }
        f32x4 s_ = {0.f, 0.f, 0.f, 0.f}, q_ = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < RED_RB; ++i) {
            if (r0 + i >= p.Mper) break;
            const size_t row = (size_t)sb * p.Mper + r0 + i;
            if (p.stats) { s_ += v[i]; q_ += v[i] * v[i]; }
            f32x4 o = v[i];
            if (p.res) o += *reinterpret_cast<const f32x4*>(p.res + row * p.ldr + c);
            *reinterpret_cast<f32x4*>(p.outF + row * p.ldf + c) = o;
        }
        if (p.stats) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int g = (c + e) / p.cpg;
                atomicAdd(&gacc[2 * g], s_[e]);
                atomicAdd(&gacc[2 * g + 1], q_[e]);
            }
        }
    }
    if (p.stats) {
        __syncthreads();
        const int G = p.N / p.cpg;
        if ((int)threadIdx.x < 2 * G) atomicAdd(&p.stats[(size_t)sb * 64 + threadIdx.x], (double)gacc[threadIdx.x]);
    }
}

GemmArgs gemm_splitk_partial_args(const GemmArgs& a) {
    GemmArgs g = a;
    g.outF = a.slab; g.ldf = a.N;
    g.bias = nullptr; g.res = nullptr; g.ldr = 0; g.outB = nullptr; g.stats = nullptr; g.cpg = 0; g.out_hw = 0; g.act = ACT_NONE;
    return g;
}
int launch_splitk_reduce(const GemmArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(a.Mper, RED_RB), a.M / a.Mper), dim3(256), 0, s, a);
    return 0;
}

template <int WM, int WN, int TM, int TN, bool CONV, int BKE = 32>
static int launch_h2_cfg(const GemmArgs& a, hipStream_t s) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    const int grid = cdiv(a.M, BM) * cdiv(a.N, BN);
    const int gy = a.splitk > 1 ? a.splitk : 1;
    size_t lds = (size_t)2 * (BM + BN) * 4 * BKE;
    if (lds < (size_t)WM * WN * 32 * 32 * TN * sizeof(float)) lds = (size_t)WM * WN * 32 * 32 * TN * sizeof(float);   // staged epilogue
    if (lds < (WM * BN * 2 + 128) * sizeof(float)) lds = (WM * BN * 2 + 128) * sizeof(float);
    static DevOnce attr_once;
    if (lds > 65536)
        UNI_LDS_OPTIN(attr_once, "gemm_h2", lds, reinterpret_cast<const void*>(&gemm_h2_kernel<WM, WN, TM, TN, CONV, true, BKE>),
                      reinterpret_cast<const void*>(&gemm_h2_kernel<WM, WN, TM, TN, CONV, false, BKE>));
    if (a.stats && gy == 1) hipLaunchKernelGGL((gemm_h2_kernel<WM, WN, TM, TN, CONV, true, BKE>), dim3(grid), dim3(64 * WM * WN), lds, s, a);
    else hipLaunchKernelGGL((gemm_h2_kernel<WM, WN, TM, TN, CONV, false, BKE>), dim3(grid, gy), dim3(64 * WM * WN), lds, s, a);
    if (gy > 1) return launch_splitk_reduce(a, s);
    return 0;
}

static bool h2_epi_ok(const GemmArgs& a) {
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    return (a.N % 8 == 0) && (!a.bias || al16(a.bias)) && (!a.res || (a.ldr % 4 == 0 && al16(a.res))) &&
           (!a.outF || (a.ldf % 4 == 0 && al16(a.outF))) && (!a.outB || a.ldb % 8 == 0);
}
// small problems (<= 1600 tiles of 64 x 64: the single-frame shapes, the stride-32 level of a batch) run on the deep-pipeline tiles of
// gemm_h2d.hip: 64 x 64 with 3 slices in flight and 3 blocks per CU up to 800 tiles, 64 x 128 (2 blocks per CU) above
// (tools/gemm_b1_bench.py, profiles/r03b_b1_gemm_tile_sweep.txt).  Returns 0 (not a deep problem) or the tile configuration.
// UNI_NO_H2D = A/B switch.
int gemm_h2d_choice(const GemmArgs& a_in) {
    static const bool off = getenv("UNI_NO_H2D") != nullptr;
    if (off || a_in.b32 != FMT_H2) return 0;
    GemmArgs a = a_in;
    a.epi = h2_epi_ok(a);
    if (!gemm_h2d_supported(a, 331)) return 0;
    const long t64 = (long)cdiv(a.M, 64) * cdiv(a.N, 64);
    // long-K plain GEMMs with hundreds of 64 x 64 tiles are bound by the L2 -> LDS path on those (4000 x 768 x 3072 = 756 tiles x 96 slices
    // x 16 KiB = 1.16 GB per launch, ~14.5 TB/s): 128 x 96 / 128 x 128 tiles move 0.6x / 0.5x the bytes, one per CU is enough when they
    // fill one round (in the model, per frame: 74 us on 256 tiles of 128 x 96 vs 85 us before, 89 us on 64 x 64)
    const bool conv = a.KH != 1 || a.KW != 1 || a.stride != 1 || a.pad != 0;
    if (!conv && a.K >= 2048 && t64 > 400 && t64 <= 800) {
        const long t96 = (long)cdiv(a.M, 128) * cdiv(a.N, 96), t128 = (long)cdiv(a.M, 128) * cdiv(a.N, 128);
        if (a.N % 96 == 0 && t96 > 192 && t96 <= 256) return 323;
        if (t128 > 192 && t128 <= 256) return 322;
    }
    if (t64 <= 800) return 331;
    if (t64 <= 1600) return (a.N % 128 == 0 || a.N % 128 > 64) ? 332 : 331;
    return 0;
}

// called by launch_gemm (gemm.hip) for FMT_H2 problems after the common argument checks
int launch_gemm_h2(const GemmArgs& a_in, hipStream_t s) {
    GemmArgs a = a_in;
    const bool conv = a.KH != 1 || a.KW != 1 || a.stride != 1 || a.pad != 0;
    UNI_REQUIRE(a.K % 8 == 0 && a.Kpad % 32 == 0 && a.lda % 8 == 0, "gemm(h2): K=%d Kpad=%d lda=%d", a.K, a.Kpad, a.lda);
    if (conv) UNI_REQUIRE(a.Cin % 8 == 0 && a.K == a.KH * a.KW * a.Cin && a.Wout < 4096 && a.Mper / a.Wout < 4096, "gemm(h2): conv K mismatch / map too large");
    if (a.stats) UNI_REQUIRE(a.cpg > 0 && 128 / a.cpg + 2 <= 64 && a.act == ACT_NONE, "gemm(h2): cpg=%d / activation with GroupNorm statistics", a.cpg);
    if (a.outB) UNI_REQUIRE(a.N % 8 == 0 && a.ldb % 8 == 0 && ((uintptr_t)a.outB & 31) == 0, "gemm(h2): operand-format output needs N, ldb multiples of 8");
    a.epi = h2_epi_ok(a);
    // tile choice: same reasoning as the bf16 kernel (largest tile that still gives >= ~1.5 blocks per CU); a step of the
    // split kernel carries 3x the MFMA work per byte, so the 256 x 256 tile pays off from fewer blocks on
    const long b44 = (long)cdiv(a.M, 256) * cdiv(a.N, 256);
    const long b22 = (long)cdiv(a.M, 128) * cdiv(a.N, 128);
    const long b21 = (long)cdiv(a.M, 128) * cdiv(a.N, 64);
    const long b12 = (long)cdiv(a.M, 64) * cdiv(a.N, 128);
    const double util44 = (double)a.N / (cdiv(a.N, 256) * 256.0);
    int cfg = a.force_cfg % 1000;
    if (a.splitk > 1) {
        UNI_REQUIRE(a.outF && a.slab && !a.outB && a.act == ACT_NONE && !a.out_hw, "gemm(h2): split-K needs fp32 output + a slab, no activation / operand-format output");
        UNI_REQUIRE(a.splitk <= 64 && a.N % 4 == 0 && a.ldf % 4 == 0 && (!a.res || a.ldr % 4 == 0) && a.Mper > 0 && a.M % a.Mper == 0 && a.epi,
                    "gemm(h2): splitk=%d N=%d ldf=%d Mper=%d", a.splitk, a.N, a.ldf, a.Mper);
        if (a.stats) UNI_REQUIRE(a.cpg > 0 && a.N % a.cpg == 0 && a.N / a.cpg <= 32, "gemm(h2): split-K statistics cpg=%d", a.cpg);
        // the persistent ping-pong kernel runs the K ranges as work units (its deep DMA stream is what small-M problems lack: the
        // generic tiles pay a full load latency per K step); everything it does not cover takes 128 x 128 (128 x 64) tiles
        // (measured, tools/gemm_b1_bench.py: implicit GEMMs do better on the 128 x 128 tiles, plain GEMMs with long K on the ping-pong kernel)
        if (cfg == 0 && gemm_h2d_choice(a)) return launch_gemm_h2d(a, 331, conv, s);      // few tiles: the K ranges are the extra blocks
        if ((cfg == 0 && !conv && a.K >= 6144) || cfg == 188) {
            GemmArgs g = gemm_splitk_partial_args(a);
            g.splitk = 0;
            const int nks = a.K / 32;
            static const bool no_q = getenv("UNI_NO_H2Q") != nullptr;
            if (!no_q && a.N > 64 && a.K % 32 == 0 && nks % a.splitk == 0 && nks / a.splitk >= 2 && gemm_h2q_supported(g)) return launch_gemm_h2q(a, s);
            UNI_REQUIRE(cfg != 188, "gemm(h2): ping-pong split-K does not support this problem");
        }
        if (gemm_h2d_has_cfg(cfg)) return launch_gemm_h2d(a, cfg, conv, s);
        if (cfg == 0 || cfg == 44 || cfg == 188) cfg = (a.N <= 64) ? 21 : 22;
    }
    if (gemm_h2d_has_cfg(cfg)) return launch_gemm_h2d(a, cfg, conv, s);
    if (cfg == 0) {
        if (a.N <= 64) cfg = (cdiv(a.M, 128) >= 256) ? 21 : 11;
        else if (util44 >= 0.74 && b44 >= 384 && a.epi) cfg = 44;
        // rounds of the persistent ping-pong kernel that are >= 70 % full beat 128 x 128 tiles at 2 blocks per CU (16000 x 1536 x 6144:
        // 378 tiles = 1.48 rounds, 794 vs 905 us)
        else if (util44 >= 0.74 && a.epi && b44 >= 180 && (double)b44 / (cdiv((int)b44, 256) * 256.0) >= 0.7 && gemm_h2q_supported(a)) cfg = 44;
        else if (b22 >= 400) cfg = 22;
        else if (conv) cfg = b12 >= 400 ? 12 : 11;
        else cfg = b21 >= 400 ? 21 : 11;
    }
    if (!a.epi && cfg == 44) cfg = 22;
    // 256-wide problems whose N is a multiple of 192 but pads the 256 x 256 tiles (N = 192, 384: a quarter of the MFMA work on zeros) or
    // leaves the last round of them mostly empty: 256 x 192 tiles of the deep kernel (8 waves, 2 slots; ~0.82 of the ping-pong kernel's
    // rate per useful flop).  Rounds-aware cost: rounds x tile area / efficiency.  UNI_NO_H2D_192 = A/B switch.
    if (a.force_cfg % 1000 == 0 && cfg == 44 && a.N % 192 == 0 && gemm_h2d_supported(a, 346)) {
        static const bool off = getenv("UNI_NO_H2D_192") != nullptr || getenv("UNI_NO_H2D") != nullptr;
        const long t192 = (long)cdiv(a.M, 256) * (a.N / 192);
        const double cost_q = (double)cdiv((int)b44, 256) * 65536.0, cost_192 = (double)cdiv((int)t192, 256) * 49152.0 / 0.82;
        if (!off && cost_192 < 0.97 * cost_q) return launch_gemm_h2d(a, 346, conv, s);
    }
    if (a.force_cfg % 1000 == 0 && cfg != 44) {
        const int d = gemm_h2d_choice(a);
        if (d) return launch_gemm_h2d(a, d, conv, s);
    }
    // problems that take the 256 x 256 tile go to the persistent ping-pong kernel (gemm_h2q.hip, cfg 188) where it covers them; UNI_NO_H2Q = A/B switch
    if (cfg == 44 && a.force_cfg % 1000 == 0) {
        static const bool no_q = getenv("UNI_NO_H2Q") != nullptr;
        if (!no_q && gemm_h2q_supported(a)) cfg = 188;
    }
    if (cfg == 188) return gemm_h2q_supported(a) ? launch_gemm_h2q(a, s) : (uni_set_error("gemm(h2): ping-pong variant does not support this problem"), -1);
#define GOH(WM, WN, TM, TN) return conv ? launch_h2_cfg<WM, WN, TM, TN, true>(a, s) : launch_h2_cfg<WM, WN, TM, TN, false>(a, s)
    switch (cfg) {
        case 44: GOH(4, 4, 2, 2);     // 256 x 256, 16 waves, 128 KiB LDS
        case 22: GOH(2, 2, 2, 2);     // 128 x 128
        case 12: GOH(2, 2, 1, 2);     // 64 x 128
        case 21: GOH(2, 2, 2, 1);     // 128 x 64
        default: GOH(2, 2, 1, 1);     // 64 x 64
    }
#undef GOH
}

// ------------------------------------------------------------------------------------------------
// host-side packing: OIHW fp32 -> [Npad][Kpad] FMT_H2 (k = (ky*KW + kx)*Cin + c), pre-scaled by `scale` (a power of two)
// ------------------------------------------------------------------------------------------------
uint16_t f32_to_f16_host(float f) {      // round-to-nearest-even, subnormals kept, overflow -> +-65504 (never inf)
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    const uint32_t mag = u & 0x7fffffffu;
    if (mag > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);                  // NaN
    if (mag >= 0x477ff000u) return (uint16_t)(sign | 0x7bffu);                 // >= 65520 rounds past the largest finite half
    if (mag < 0x33000001u) return (uint16_t)sign;                              // <= 2^-25: rounds to zero
    const int e = (int)(mag >> 23) - 127;
    uint32_t man = (mag & 0x7fffffu) | 0x800000u;                              // 24-bit significand
    int shift;                                                                 // bits to drop
    uint32_t base;
    if (e < -14) { shift = 13 + (-14 - e); base = 0; }                         // subnormal half: value = man * 2^(e-23) / 2^-24
    else { shift = 13; base = (uint32_t)(e + 15) << 10; man &= 0x7fffffu; }
    uint32_t q = man >> shift;
    const uint32_t rem = man & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (q & 1u))) ++q;                    // RNE (a carry out of the mantissa bumps the exponent: correct)
    return (uint16_t)(sign | (base + q));
}
float f16_to_f32_host(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const int e = (h >> 10) & 31;
    const uint32_t m = h & 0x3ffu;
    float v;
    if (e == 0) v = (float)m * 5.9604644775390625e-8f;                         // m * 2^-24
    else if (e == 31) v = m ? NAN : INFINITY;
    else { uint32_t u = ((uint32_t)(e + 112) << 23) | (m << 13); memcpy(&v, &u, 4); }
    uint32_t u;
    memcpy(&u, &v, 4);
    u |= sign;
    memcpy(&v, &u, 4);
    return v;
}
void pack_weight_h2_host(const float* w, int N, int Cin, int KH, int KW, const float* row_scale, float scale, uint16_t* out,
                         int Npad, int Kpad) {
    const int K = KH * KW * Cin;
    memset(out, 0, (size_t)Npad * Kpad * 4);
    for (int n = 0; n < N; ++n) {
        const float sc = (row_scale ? row_scale[n] : 1.f);
        uint16_t* o = out + (size_t)n * Kpad * 2;
        const float* wn = w + (size_t)n * K;
        for (int c = 0; c < Cin; ++c)
            for (int ky = 0; ky < KH; ++ky)
                for (int kx = 0; kx < KW; ++kx) {
                    const int k = (ky * KW + kx) * Cin + c;
                    const float x = sc * wn[(c * KH + ky) * KW + kx] * scale;
                    const uint16_t hi = f32_to_f16_host(x);
                    const uint16_t lo = f32_to_f16_host(x - f16_to_f32_host(hi));
                    o[(k >> 3) * 16 + (k & 7)] = hi;
                    o[(k >> 3) * 16 + 8 + (k & 7)] = lo;
                }
    }
}
// power-of-two scale that puts max |w| just below 2^15 (hi halves finite, lo halves of all but negligible weights normal)
float h2_weight_scale(float maxabs) {
    if (!(maxabs > 0.f) || !std::isfinite(maxabs)) return 1.f;
    int e;
    (void)frexpf(maxabs, &e);               // maxabs = f * 2^e, f in [0.5, 1)
    return ldexpf(1.f, 15 - e);             // maxabs * scale in [2^14, 2^15)
}
