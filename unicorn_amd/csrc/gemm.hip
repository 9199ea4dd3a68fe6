// K1b / K2 / K5: 16-bit MFMA implicit-GEMM for every dense contraction of the path
// (ConvNeXt pointwise MLPs, PAFPN/head 1x1 and 3x3 convs, 2x2/s2 downsample, transformer Linears).
//
//   out[m][n] = epilogue( sum_k A[m][k] * W[n][k] )        m = output pixel, n = out channel
//
// A is an NHWC bf16 activation map; for CONV the k index is (ky,kx,c) and the A tile is gathered on the
// fly with zero padding (implicit GEMM, nothing is materialised).  W is pre-packed [Npad][Kpad] bf16
// (K contiguous, zero padded).  fp32 accumulation on v_mfma_f32_32x32x16_bf16.
//
// CDNA4 design notes
//  * 256 threads = 4 waves (2x2); wave tile (32*TM pixels) x (32*TN channels); block tile BM x BN =
//    (64*TM) x (64*TN); BK = 64.
//  * global -> LDS goes through the LDS-DMA path (global_load_lds_dwordx4, 1 KiB per wave instruction, no
//    VGPR round trip), two LDS buffers, ONE barrier per K step: the loads of tile k+1 are in flight while
//    tile k is multiplied.  The DMA writes lane-linear (base + lane*16 B), so the bank-conflict XOR swizzle
//    chunk' = chunk ^ ((row>>1)&7) is applied on the per-lane SOURCE address and again on the ds_read_b128
//    fragment reads (both-sides-or-neither).  Out-of-image / K-tail lanes read a 16-byte zero page.
//  * The MFMA is issued "swapped": weights are the A operand, pixels the B operand, so every lane ends up
//    with 4 CONSECUTIVE output channels of one pixel per accumulator quad.  The epilogue is therefore fully
//    vectorised (float4 bias / residual / fp32 stores) and bf16 outputs are widened to 16-byte stores with
//    v_permlane32_swap (store-issue count, not bandwidth, bounds small-K GEMMs on this chip).
//  * GroupNorm statistics of the raw conv output are reduced in the epilogue (reduce-scatter butterfly over
//    the 32 pixel lanes -> LDS -> one double atomic per group per block).
//  * Blocks are remapped so each XCD (private 4 MiB L2) owns a contiguous range of M panels.
#include "kernels.h"
#include <cstdlib>

__device__ u32x4 g_zero_page = {0u, 0u, 0u, 0u};

#define GLDS16(gptr, lptr)                                                                             \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),            \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

#define OPAQUE64(x)                                      \
    do {                                                 \
        int _lo = (int)(x), _hi = (int)((x) >> 32);      \
        asm volatile("" : "+v"(_lo), "+v"(_hi));         \
        (x) = ((long)_hi << 32) | (unsigned)_lo;         \
    } while (0)

#include "gemm_epi.h"

template <int WM, int WN, int TM, int TN, int BK, bool CONV, bool STATS, int NSTG = 2>
__global__ __launch_bounds__(64 * WM * WN) void gemm_bf16_kernel(GemmArgs p) {
    constexpr int NW = WM * WN;                                  // waves per block, arranged WM (pixels) x WN (channels)
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int CPR = BK / 8;             // 16-B chunks per LDS row (row = BK bf16 = 128 B or 64 B)
    constexpr int RPP = 64 / CPR;           // rows per 1-KiB DMA piece (8 or 16)
    constexpr int SW = BK == 64 ? 1 : 2;    // swizzle: chunk' = chunk ^ ((row >> SW) & (CPR-1))  (256-B bank period)
    constexpr int A_PC = BM / RPP / NW;     // 1-KiB pieces per wave for the A tile
    constexpr int B_PC = BN / RPP / NW;
    static_assert(A_PC >= 1 && B_PC >= 1 && NW % 2 == 0, "tile too small for the wave grid");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* As = reinterpret_cast<bf16*>(smem);                 // [NSTG][BM*BK]
    bf16* Bs = As + NSTG * BM * BK;                           // [NSTG][BN*BK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // ---- XCD-aware bijective block remap (block b runs on XCD b%8) ----
    const int nbn = (p.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int L;
    {
        const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    // L2-friendly tile order: N is cut into chunks of 8 tiles; inside a chunk tiles run M-major, so the ~64 blocks
    // resident on one XCD form an (8 M-panels x 8 N-tiles) super-tile that shares 16 operand K-slices per step.
    int bm, bn;
    {
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
    if ((p.dbg & 64) && (blockIdx.x & 8) && blockIdx.x < 256) {   // experiment: de-phase half of the first-round blocks
        for (int z = 0; z < (p.dbg >> 8); ++z) __builtin_amdgcn_s_sleep(127);
    }

    // ---- per-lane DMA source descriptors ----
    // piece pc = wave + NW*i covers tile rows RPP*pc .. RPP*pc+RPP-1; lane -> row RPP*pc + lane/CPR, physical chunk
    // lane%CPR; logical chunk = physical ^ swizzle(row), which is independent of i for both BK (see header).
    const int lrow = lane / CPR;
    const int lch = (lane % CPR) ^ (((RPP * wave + lrow) >> SW) & (CPR - 1));
    const char* abase = reinterpret_cast<const char*>(p.A);
    const long zoff = reinterpret_cast<const char*>(&g_zero_page) - abase;
    int a_pix[A_PC];
#pragma unroll
    for (int i = 0; i < A_PC; ++i) {
        int m = m0 + RPP * (wave + NW * i) + lrow;
        m = m < p.M ? m : p.M - 1;
        if (CONV) {
            int b = m / p.Mper, q = m - b * p.Mper;        // sample, pixel within the sample
            int oy = q / p.Wout, ox = q - oy * p.Wout;
            a_pix[i] = (b << 24) | (oy << 12) | ox;
        } else {
            a_pix[i] = m;
        }
    }
    const bf16* wbase = p.W + (size_t)(n0 + RPP * wave + lrow) * p.Kpad + lch * 8;

    auto issue = [&](int kt, int buf) {
        const int k = kt * BK + lch * 8;
        const bool kok = k < p.K;
        char* adst = reinterpret_cast<char*>(As + buf * BM * BK) + wave * 1024;
        char* bdst = reinterpret_cast<char*>(Bs + buf * BN * BK) + wave * 1024;
        if (CONV) {
            int tap = k / p.Cin;
            int c = k - tap * p.Cin;
            int ky = tap / p.KW, kx = tap - ky * p.KW;
#pragma unroll
            for (int i = 0; i < A_PC; ++i) {
                int bb = a_pix[i] >> 24, oy = (a_pix[i] >> 12) & 0xfff, ox = a_pix[i] & 0xfff;
                int iy = oy * p.stride - p.pad + ky, ix = ox * p.stride - p.pad + kx;
                bool ok = kok && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
                long off = ok ? (long)(((size_t)((bb * p.Hin + iy) * p.Win + ix) * p.lda + c) * sizeof(bf16)) : zoff;
                OPAQUE64(off);   // keep ONE DMA per piece (hipcc otherwise splits the select into exec-masked branches)
                GLDS16(abase + off, adst + i * NW * 1024);
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_PC; ++i) {
                long off = kok ? (long)(((size_t)a_pix[i] * p.lda + k) * sizeof(bf16)) : zoff;
                OPAQUE64(off);
                GLDS16(abase + off, adst + i * NW * 1024);
            }
        }
#pragma unroll
        for (int i = 0; i < B_PC; ++i) GLDS16(wbase + (size_t)i * NW * RPP * p.Kpad + kt * BK, bdst + i * NW * 1024);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.Kpad / BK;
    const int fr = lane & 31;          // fragment row within a 32-row sub-tile
    const int fh = lane >> 5;          // which 8-wide half of the 16-deep MFMA K step
    // NSTG == 2: one tile in flight, __syncthreads (vmcnt(0) + barrier) per K step.
    // NSTG  > 2: NSTG-1 tiles in flight, COUNTED vmcnt (never 0 in steady state) + raw s_barrier: the K loop is bound by the
    // global->LDS round trip (~3k cycles under load), so bandwidth scales with the bytes in flight.
    constexpr int DEPTH = NSTG - 1, PCW = A_PC + B_PC;
    if (NSTG > 2) {
        for (int s_ = 0; s_ < DEPTH && s_ < nk; ++s_) issue(s_, s_);
    } else {
        issue(0, 0);
    }
    for (int kt = 0; kt < nk; ++kt) {
        int buf;
        if (NSTG > 2) {
            buf = kt % NSTG;
            const int rem = nk - 1 - kt;             // stages after this one
            const int fly = rem < DEPTH - 1 ? rem : DEPTH - 1;   // stages allowed to stay in flight
            if (fly >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PCW) : "memory");
            else if (fly == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PCW) : "memory");
            else if (fly == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * PCW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();            // stage kt landed for every wave; everyone is done with stage kt-1
            asm volatile("" ::: "memory");
            if (kt + DEPTH < nk) issue(kt + DEPTH, (kt + DEPTH) % NSTG);
        } else {
            buf = kt & 1;
            if (!(p.dbg & 8)) __syncthreads();           // tile kt landed (vmcnt(0) + barrier); everyone is done with buf^1
            if (kt + 1 < nk && !(p.dbg & 1)) issue(kt + 1, buf ^ 1);
        }
        const bf16* a = As + buf * BM * BK;
        const bf16* b = Bs + buf * BN * BK;
        // software-pipelined fragment reads: the ds_read_b128s of sub-step kk+1 are in flight while the MFMAs of
        // sub-step kk issue (hipcc does not pipeline them across the unrolled kk loop by itself)
        bf16x8 fa[2][TM], fb[2][TN];
        auto ldfrag = [&](int kk, int slot) {
            const int ch = kk * 2 + fh;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                int row = wm * 32 * TM + i * 32 + fr;
                fa[slot][i] = *reinterpret_cast<const bf16x8*>(a + row * BK + ((ch ^ ((row >> SW) & (CPR - 1))) << 3));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                int row = wn * 32 * TN + j * 32 + fr;
                fb[slot][j] = *reinterpret_cast<const bf16x8*>(b + row * BK + ((ch ^ ((row >> SW) & (CPR - 1))) << 3));
            }
        };
        if (p.dbg & 2) continue;   // ablation: DMA only
        if (!(p.dbg & 4) || kt == 0) ldfrag(0, 0);
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            if (kk + 1 < BK / 16 && (!(p.dbg & 4) || kt == 0)) ldfrag(kk + 1, (kk + 1) & 1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)   // swapped: D[n][m], lane -> pixel m = lane&31, channels 4*fh + (r&3) + 8*(r>>2)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kk & 1][j], fa[kk & 1][i], acc[i][j], 0, 0, 0);
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
    gemm_epilogue<WM, WN, TM, TN, STATS>(p, acc, m0, n0, wm, wn, lane, tid, smem);
}

// ------------------------------------------------------------------------------------------------
// Exact-fp32 variant (precision mode "fp32"): same LDS-DMA / swizzle / epilogue, operands are fp32 and the
// contraction runs on v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain, 157 TF peak).  64x64 block tile, BK = 32
// floats (= the same 128-byte LDS rows).  K-permutation: MFMA step t of a K slab contracts dims {t, 16+t}.
// ------------------------------------------------------------------------------------------------
template <bool CONV>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs p) {
    constexpr int BM = 64, BN = 64, BKF = 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);               // [2][BM*BKF]
    float* Bs = As + 2 * BM * BKF;                            // [2][BN*BKF]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nbn = (p.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int L;
    {
        const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int bm = L / nbn, bn = L % nbn;
    const int m0 = bm * BM, n0 = bn * BN;
    const int lrow = lane >> 3;
    const int lch = (lane & 7) ^ ((4 * wave + (lane >> 4)) & 7);
    const float* A = reinterpret_cast<const float*>(p.A);
    const float* W = reinterpret_cast<const float*>(p.W);
    const char* abase = reinterpret_cast<const char*>(A);
    const long zoff = reinterpret_cast<const char*>(&g_zero_page) - abase;
    int a_pix[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int m = m0 + 8 * (wave + 4 * i) + lrow;
        m = m < p.M ? m : p.M - 1;
        if (CONV) {
            int b = m / p.Mper, q = m - b * p.Mper;        // sample, pixel within the sample
            int oy = q / p.Wout, ox = q - oy * p.Wout;
            a_pix[i] = (b << 24) | (oy << 12) | ox;
        } else {
            a_pix[i] = m;
        }
    }
    const float* wbase = W + (size_t)(n0 + 8 * wave + lrow) * p.Kpad + lch * 4;
    auto issue = [&](int kt, int buf) {
        const int k = kt * BKF + lch * 4;
        const bool kok = k < p.K;
        char* adst = reinterpret_cast<char*>(As + buf * BM * BKF) + wave * 1024;
        char* bdst = reinterpret_cast<char*>(Bs + buf * BN * BKF) + wave * 1024;
        int ky = 0, kx = 0, c = k;
        if (CONV) {
            int tap = k / p.Cin;
            c = k - tap * p.Cin;
            ky = tap / p.KW;
            kx = tap - ky * p.KW;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            long off;
            if (CONV) {
                int bb = a_pix[i] >> 24, oy = (a_pix[i] >> 12) & 0xfff, ox = a_pix[i] & 0xfff;
                int iy = oy * p.stride - p.pad + ky, ix = ox * p.stride - p.pad + kx;
                bool ok = kok && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
                off = ok ? (long)(((size_t)((bb * p.Hin + iy) * p.Win + ix) * p.lda + c) * sizeof(float)) : zoff;
            } else {
                off = kok ? (long)(((size_t)a_pix[i] * p.lda + k) * sizeof(float)) : zoff;
            }
            OPAQUE64(off);
            GLDS16(abase + off, adst + i * 4096);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) GLDS16(wbase + (size_t)i * 32 * p.Kpad + kt * BKF, bdst + i * 4096);
    };
    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
    const int nk = p.Kpad / BKF;
    const int fr = lane & 31, fh = lane >> 5;
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        __syncthreads();
        if (kt + 1 < nk) issue(kt + 1, buf ^ 1);
        const int arow = wm * 32 + fr, brow = wn * 32 + fr;
        const float* a = As + buf * BM * BKF + arow * BKF;
        const float* b = Bs + buf * BN * BKF + brow * BKF;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch = 4 * fh + j;
            f32x4 fa = *reinterpret_cast<const f32x4*>(a + ((ch ^ ((arow >> 1) & 7)) << 2));
            f32x4 fb = *reinterpret_cast<const f32x4*>(b + ((ch ^ ((brow >> 1) & 7)) << 2));
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[e], fa[e], acc[0][0], 0, 0, 0);
        }
    }
    if (p.stats) gemm_epilogue<2, 2, 1, 1, true>(p, acc, m0, n0, wm, wn, lane, tid, smem);
    else gemm_epilogue<2, 2, 1, 1, false>(p, acc, m0, n0, wm, wn, lane, tid, smem);
}

template <int WM, int WN, int TM, int TN, int BK, bool CONV, int NSTG = 2>
static int launch_cfg(const GemmArgs& a, hipStream_t s) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    int grid = cdiv(a.M, BM) * cdiv(a.N, BN);
    size_t lds = (size_t)NSTG * (BM + BN) * BK * sizeof(bf16);
    if (lds < (size_t)WM * WN * 32 * 32 * TN * sizeof(float)) lds = (size_t)WM * WN * 32 * 32 * TN * sizeof(float);   // staged epilogue
    if (lds < (WM * BN * 2 + 128) * sizeof(float)) lds = (WM * BN * 2 + 128) * sizeof(float);
    static DevOnce attr_once;      // > 64 KiB dynamic LDS needs the opt-in attribute
    if (lds > 65536)
        UNI_LDS_OPTIN(attr_once, "gemm_bf16", lds, reinterpret_cast<const void*>(&gemm_bf16_kernel<WM, WN, TM, TN, BK, CONV, true, NSTG>),
                      reinterpret_cast<const void*>(&gemm_bf16_kernel<WM, WN, TM, TN, BK, CONV, false, NSTG>));
    if (a.stats) hipLaunchKernelGGL((gemm_bf16_kernel<WM, WN, TM, TN, BK, CONV, true, NSTG>), dim3(grid), dim3(64 * WM * WN), lds, s, a);
    else hipLaunchKernelGGL((gemm_bf16_kernel<WM, WN, TM, TN, BK, CONV, false, NSTG>), dim3(grid), dim3(64 * WM * WN), lds, s, a);
    return 0;
}

int launch_gemm(const GemmArgs& a_in, hipStream_t s) {
    GemmArgs a = a_in;
    if (a.Mper <= 0) a.Mper = a.M;                       // single sample
    UNI_REQUIRE(a.M % a.Mper == 0 && a.M / a.Mper < 128, "gemm: M=%d is not a multiple of Mper=%d", a.M, a.Mper);
    UNI_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
    UNI_REQUIRE(((uintptr_t)a.A & 15) == 0, "gemm: A must be 16-byte aligned");
    if (a.b32 == FMT_H2) return launch_gemm_h2(a, s);
    UNI_REQUIRE(a.K % (a.b32 == FMT_F32 ? 4 : 8) == 0 && a.Kpad % 64 == 0 && a.Kpad >= a.K, "gemm: K=%d Kpad=%d", a.K, a.Kpad);
    UNI_REQUIRE(a.lda % (a.b32 == FMT_F32 ? 4 : 8) == 0 && ((uintptr_t)a.A & 15) == 0, "gemm: lda=%d / A must be 16-byte aligned", a.lda);
    const bool conv = a.KH != 1 || a.KW != 1 || a.stride != 1 || a.pad != 0;
    if (a.b32 == FMT_F32) {
        if (conv) UNI_REQUIRE(a.Cin % 4 == 0 && a.K == a.KH * a.KW * a.Cin && a.Wout < 4096 && a.Mper / a.Wout < 4096, "gemm(f32): conv K mismatch / map too large");
        if (a.stats) UNI_REQUIRE(a.cpg > 0 && 128 / a.cpg + 2 <= 64, "gemm: cpg=%d unsupported", a.cpg);
        const int grid = cdiv(a.M, 64) * cdiv(a.N, 64);
        const size_t lds = (size_t)2 * (64 + 64) * 32 * sizeof(float);
        if (conv) hipLaunchKernelGGL((gemm_f32_kernel<true>), dim3(grid), dim3(256), lds, s, a);
        else hipLaunchKernelGGL((gemm_f32_kernel<false>), dim3(grid), dim3(256), lds, s, a);
        return 0;
    }
    if (conv) UNI_REQUIRE(a.Cin % 8 == 0 && a.K == a.KH * a.KW * a.Cin && a.Wout < 4096 && a.Mper / a.Wout < 4096, "gemm: conv K mismatch / map too large");
    if (a.stats) UNI_REQUIRE(a.cpg > 0 && 128 / a.cpg + 2 <= 64, "gemm: cpg=%d unsupported", a.cpg);
    if (a.stats) UNI_REQUIRE(a.act == ACT_NONE, "gemm: GroupNorm statistics and an epilogue activation are exclusive");
    // tile choice (measured on MI355X, tools/gemm_bench.py at batch-1 and batch-8 row counts): the K loop is bound by
    // the global->LDS fill rate, so the biggest block tile that still yields >= ~1.5 blocks per CU wins (256x256,
    // 16 waves); small problems want >= ~400 blocks of a smaller tile; implicit convs amortise their gather
    // address math over wider-N tiles.
    {   // LDS-staged row-coalesced stores need vector-aligned operands; odd widths (cls/obj/reg heads) keep the direct path
        auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
        a.epi = (a.N % 8 == 0) && (!a.bias || al16(a.bias)) && (!a.res || (a.ldr % 4 == 0 && al16(a.res))) &&
                (!a.outF || (a.ldf % 4 == 0 && al16(a.outF))) && (!a.outB || (a.ldb % 8 == 0 && al16(a.outB))) &&
                !((a.force_cfg / 1000) & 32);
    }
    static const bool no44 = getenv("UNI_NO44") != nullptr, no_p44 = getenv("UNI_NO_P44") != nullptr;   // A/B switches (read once)
    const long b44 = (long)cdiv(a.M, 256) * cdiv(a.N, 256);
    const long b22 = (long)cdiv(a.M, 128) * cdiv(a.N, 128);
    const long b21 = (long)cdiv(a.M, 128) * cdiv(a.N, 64);
    const long b12 = (long)cdiv(a.M, 64) * cdiv(a.N, 128);
    const double util44 = (double)a.N / (cdiv(a.N, 256) * 256.0);
    int cfg = a.force_cfg % 1000;
    if (cfg == 0) {
        if (a.N <= 64) cfg = (cdiv(a.M, 128) >= 256) ? 21 : 11;
        else if (util44 >= 0.74 && (b44 >= 1500 || (b44 >= 384 && a.K >= 512)) && !no44) cfg = 44;
        else if (b22 >= 512) cfg = 22;
        else if (conv) cfg = b22 >= 400 ? 22 : (b12 >= 400 ? 12 : 11);
        else cfg = b21 >= 400 ? 21 : 11;
    }
#define GO(WM, WN, TM, TN, BK) return conv ? launch_cfg<WM, WN, TM, TN, BK, true>(a, s) : launch_cfg<WM, WN, TM, TN, BK, false>(a, s)
    // 256-wide problems with whole rounds of tiles go to the persistent ping-pong kernel (gemm_h2q.hip, the bf16 instantiations;
    // 188 forces it, UNI_NO_H2Q = A/B switch back to the kernels below)
    {
        static const bool no_q = getenv("UNI_NO_H2Q") != nullptr;
        const bool rounds = b44 >= 384 || (b44 >= 180 && (double)b44 / (cdiv((int)b44, 256) * 256.0) >= 0.7);
        // measured per shape against the kernels below (profiles/r02e_bf16_gemm_shapes.txt): with ONE MFMA per product the 8-MFMA phases are
        // shorter than the LDS / DMA phases, so the schedule only wins where the K loop is long and the A operand streams plainly:
        // plain GEMMs with K >= 1024 (-5..-15 %); implicit GEMMs (+3..10 %) and K <= 768 layers (0..+8 %) stay where they were
        const bool pays = !conv && !a.stats && a.K >= 1024;
        if (cfg == 188 || (a.force_cfg % 1000 == 0 && !no_q && pays && a.N > 64 && util44 >= 0.74 && rounds && a.epi && gemm_h2q_supported(a))) {
            if (!gemm_h2q_supported(a)) { uni_set_error("gemm: ping-pong variant does not support this problem"); return -1; }
            return launch_gemm_h2q(a, s);
        }
    }
    // plain GEMMs that would take the 256x256 tile go to the persistent variant (gemm_p44.hip): -5..-20 % on the MLP shapes
    if (cfg == 44 && a.force_cfg % 1000 == 0 && gemm_p44_supported(a) && !no_p44) cfg = 144;
    if (cfg == 144 && !gemm_p44_supported(a)) cfg = 44;
    if (cfg == 144) return launch_gemm_p44(a, s);
    if (!a.epi && (cfg == 44 || cfg == 42 || cfg == 24)) cfg = 22;
#define GOS(WM, WN, TM, TN, BK, NS) return conv ? launch_cfg<WM, WN, TM, TN, BK, true, NS>(a, s) : launch_cfg<WM, WN, TM, TN, BK, false, NS>(a, s)
    if (!a.epi && (cfg == 444 || cfg == 445)) cfg = 22;
    switch (cfg) {
        case 444: GOS(4, 4, 2, 2, 32, 4);   // 256 x 256, BK = 32, 4 stages (3 in flight, 96 KiB), 128 KiB LDS
        case 445: GOS(4, 4, 2, 2, 32, 5);   // 5 stages (4 in flight, 128 KiB), 160 KiB LDS
        case 224: GOS(2, 2, 2, 2, 32, 4);   // 128 x 128, BK = 32, 4 stages (64 KiB LDS like cfg 22, 48 KiB in flight)   // 2-digit codes: 4 waves (2x2), wave tile 32TM x 32TN; 44 / 42 / 24: 16 / 8 / 8 waves of 64x64 wave tiles
        case 44: GO(4, 4, 2, 2, 64);     // 256 x 256 block tile, 128 KiB LDS
        case 42: GO(4, 2, 2, 2, 64);     // 256 x 128
        case 24: GO(2, 4, 2, 2, 64);     // 128 x 256
        case 122: GO(2, 2, 2, 2, 32);
        case 22: GO(2, 2, 2, 2, 64);
        case 12: GO(2, 2, 1, 2, 64);
        case 21: GO(2, 2, 2, 1, 64);
        default: GO(2, 2, 1, 1, 64);
    }
#undef GO
#undef GOS
}
