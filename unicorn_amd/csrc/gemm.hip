// K1b / K2 / K5: bf16 MFMA implicit-GEMM for every dense contraction of the path
// (ConvNeXt pointwise MLPs, PAFPN/head 1x1 and 3x3 convs, 2x2/s2 downsample, transformer Linears).
//
//   out[m][n] = epilogue( sum_k A[m][k] * W[n][k] )        m = output pixel, n = out channel
//
// A is an NHWC bf16 activation map; for CONV the k index is (ky,kx,c) and the A tile is gathered
// on the fly with zero padding (implicit GEMM, nothing is materialised). W is pre-packed
// [Npad][Kpad] bf16 (K contiguous, zero padded).  fp32 accumulate on v_mfma_f32_32x32x16_bf16.
//
// Geometry: 256 threads = 4 waves (2x2), wave tile = (32*TM) x (32*TN), block tile BM x BN =
// (64*TM) x (64*TN), BK = 64.  Register-staged global->LDS (needed for the zero-padded gather),
// two LDS buffers, one barrier per K step.  LDS rows are 128 B; 16-B chunks are XOR-swizzled with
// ((row>>1)&7) so that the ds_read_b128 fragment reads of any 16-lane group hit 16 distinct
// (row-parity, chunk) bank slots (MI355X LDS: 64 banks x 4 B, b128 groups of 16 lanes).
// Blocks are remapped so each XCD (private L2) owns a contiguous range of M panels.
#include "kernels.h"

template <int A_CH, int B_CH, bool CONV>
__device__ __forceinline__ void gemm_load_tiles(const GemmArgs& p, int kt, int kc, const int (&a_pix)[A_CH],
                                                const bf16* wbase, u32x4 (&ra)[A_CH], u32x4 (&rb)[B_CH]) {
    const int k = kt * 64 + kc * 8;
    const bool kok = k < p.K;
    if (CONV) {
        int tap = k / p.Cin;
        int c = k - tap * p.Cin;
        int ky = tap / p.KW, kx = tap - ky * p.KW;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            int oy = a_pix[i] >> 16, ox = a_pix[i] & 0xffff;
            int iy = oy * p.stride - p.pad + ky, ix = ox * p.stride - p.pad + kx;
            bool ok = kok && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
            const bf16* src = p.A + ((size_t)(iy * p.Win + ix) * p.lda + c);
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ok) v = *reinterpret_cast<const u32x4*>(src);
            ra[i] = v;
        }
    } else {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const bf16* src = p.A + ((size_t)a_pix[i] * p.lda + k);
            u32x4 v = {0u, 0u, 0u, 0u};
            if (kok) v = *reinterpret_cast<const u32x4*>(src);
            ra[i] = v;
        }
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i)
        rb[i] = *reinterpret_cast<const u32x4*>(wbase + (size_t)i * 32 * p.Kpad + kt * 64);
}

template <int A_CH, int B_CH>
__device__ __forceinline__ void gemm_store_tiles(bf16* a, bf16* b, int lrow, int kc, const u32x4 (&ra)[A_CH],
                                                 const u32x4 (&rb)[B_CH]) {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        int row = lrow + i * 32;
        *reinterpret_cast<u32x4*>(a + row * 64 + ((kc ^ ((row >> 1) & 7)) << 3)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
        int row = lrow + i * 32;
        *reinterpret_cast<u32x4*>(b + row * 64 + ((kc ^ ((row >> 1) & 7)) << 3)) = rb[i];
    }
}

template <int TM, int TN, bool CONV>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs p) {
    constexpr int BM = 64 * TM, BN = 64 * TN, BK = 64;
    constexpr int A_CH = BM / 32;   // 16-B chunks per thread for the A tile
    constexpr int B_CH = BN / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* As = reinterpret_cast<bf16*>(smem);                 // [2][BM*BK]
    bf16* Bs = As + 2 * BM * BK;                              // [2][BN*BK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- XCD-aware bijective block remap (block b runs on XCD b%8) ----
    const int nbn = (p.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int L;
    {
        const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int bm = L / nbn, bn = L % nbn;
    const int m0 = bm * BM, n0 = bn * BN;

    // ---- per-thread load descriptors ----
    const int kc = tid & 7;            // 16-B chunk within the 64-wide K slab
    const int lrow = tid >> 3;         // 0..31
    int a_pix[A_CH];                   // non-conv: element offset of row start; conv: packed (oy<<16|ox)
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        int m = m0 + lrow + i * 32;
        m = m < p.M ? m : p.M - 1;
        if (CONV) {
            int oy = m / p.Wout, ox = m - oy * p.Wout;
            a_pix[i] = (oy << 16) | ox;
        } else {
            a_pix[i] = m;
        }
    }
    const bf16* wbase = p.W + (size_t)(n0 + lrow) * p.Kpad + kc * 8;

    u32x4 ra[A_CH], rb[B_CH];
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.Kpad / BK;
    gemm_load_tiles<A_CH, B_CH, CONV>(p, 0, kc, a_pix, wbase, ra, rb);
    gemm_store_tiles<A_CH, B_CH>(As, Bs, lrow, kc, ra, rb);
    __syncthreads();

    const int fr = lane & 31;          // fragment row (A) / col (B)
    const int fh = lane >> 5;          // which 8-wide half of the 16-deep MFMA K step
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gemm_load_tiles<A_CH, B_CH, CONV>(p, kt + 1, kc, a_pix, wbase, ra, rb);
        const bf16* a = As + buf * BM * BK;
        const bf16* b = Bs + buf * BN * BK;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 fa[TM], fb[TN];
            const int ch = kk * 2 + fh;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                int row = wm * 32 * TM + i * 32 + fr;
                fa[i] = *reinterpret_cast<const bf16x8*>(a + row * BK + ((ch ^ ((row >> 1) & 7)) << 3));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                int row = wn * 32 * TN + j * 32 + fr;
                fb[j] = *reinterpret_cast<const bf16x8*>(b + row * BK + ((ch ^ ((row >> 1) & 7)) << 3));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) gemm_store_tiles<A_CH, B_CH>(As + (buf ^ 1) * BM * BK, Bs + (buf ^ 1) * BN * BK, lrow, kc, ra, rb);
        __syncthreads();
    }

    // ---- epilogue ----
    // C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    float gsum[TN], gsq[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { gsum[j] = 0.f; gsq[j] = 0.f; }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * 32 * TN + j * 32 + fr;
        const bool cok = col < p.N;
        const float bias = (p.bias && cok) ? p.bias[col] : 0.f;
        const int act = (col >= p.act_col0) ? p.act : ACT_NONE;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (row < p.M && cok) {
                    float v = acc[i][j][r] + bias;
                    gsum[j] += v;
                    gsq[j] += v * v;
                    v = act_apply(v, act);
                    if (p.res) v += p.res[(size_t)row * p.ldr + col];
                    if (p.outF) p.outF[(size_t)row * p.ldf + col] = v;
                    if (p.outB) p.outB[(size_t)row * p.ldb + col] = (bf16)v;
                }
            }
        }
    }
    if (p.stats) {
        // per-column partial sums -> per-GroupNorm-group sums -> one double atomic per group per block
        float* red = reinterpret_cast<float*>(smem);          // [2 wm][BN][2]
        float* gacc = red + 2 * BN * 2;                       // [64][2]
        __syncthreads();
        if (tid < 128) gacc[tid] = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s = gsum[j] + __shfl_xor(gsum[j], 32, 64);
            float q = gsq[j] + __shfl_xor(gsq[j], 32, 64);
            if (fh == 0) {
                int c = wn * 32 * TN + j * 32 + fr;
                red[(wm * BN + c) * 2 + 0] = s;
                red[(wm * BN + c) * 2 + 1] = q;
            }
        }
        __syncthreads();
        const int g_first = n0 / p.cpg;
        if (tid < BN && n0 + tid < p.N) {
            float s = red[tid * 2] + red[(BN + tid) * 2];
            float q = red[tid * 2 + 1] + red[(BN + tid) * 2 + 1];
            int gl = (n0 + tid) / p.cpg - g_first;
            atomicAdd(&gacc[gl * 2], s);
            atomicAdd(&gacc[gl * 2 + 1], q);
        }
        __syncthreads();
        const int nloc = (min(n0 + BN, p.N) - 1) / p.cpg - g_first + 1;
        if (tid < nloc * 2) atomicAdd(&p.stats[(g_first + (tid >> 1)) * 2 + (tid & 1)], (double)gacc[tid]);
    }
}

template <int TM, int TN, bool CONV>
static int launch_cfg(const GemmArgs& a, hipStream_t s) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    int grid = cdiv(a.M, BM) * cdiv(a.N, BN);
    size_t lds = (size_t)2 * (BM + BN) * 64 * sizeof(bf16);
    if (lds < (2 * BN * 2 + 128) * sizeof(float)) lds = (2 * BN * 2 + 128) * sizeof(float);
    hipLaunchKernelGGL((gemm_bf16_kernel<TM, TN, CONV>), dim3(grid), dim3(256), lds, s, a);
    return 0;
}

int launch_gemm(const GemmArgs& a, hipStream_t s) {
    UNI_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
    UNI_REQUIRE(a.K % 8 == 0 && a.Kpad % 64 == 0 && a.Kpad >= a.K, "gemm: K=%d Kpad=%d", a.K, a.Kpad);
    UNI_REQUIRE(a.lda % 8 == 0, "gemm: lda=%d must be a multiple of 8", a.lda);
    const bool conv = a.KH != 1 || a.KW != 1 || a.stride != 1 || a.pad != 0;
    if (conv) UNI_REQUIRE(a.Cin % 8 == 0 && a.K == a.KH * a.KW * a.Cin, "gemm: conv K mismatch");
    if (a.stats) UNI_REQUIRE(a.cpg > 0 && 128 / a.cpg + 2 <= 64, "gemm: cpg=%d unsupported", a.cpg);
    // tile choice: fill >= ~2 waves of blocks on 256 CUs, prefer the big tile (Npad is a multiple of 128)
    const long b22 = (long)cdiv(a.M, 128) * cdiv(a.N, 128);
    const long b12 = (long)cdiv(a.M, 64) * cdiv(a.N, 128);
    int cfg = a.force_cfg;
    if (cfg == 0) {
        if (a.N <= 64) cfg = (cdiv(a.M, 128) >= 256) ? 21 : 11;
        else if (b22 >= 384) cfg = 22;
        else if (b12 >= 256) cfg = 12;
        else cfg = 11;
    }
#define GO(TM, TN) return conv ? launch_cfg<TM, TN, true>(a, s) : launch_cfg<TM, TN, false>(a, s)
    switch (cfg) {
        case 22: GO(2, 2);
        case 12: GO(1, 2);
        case 21: GO(2, 1);
        default: GO(1, 1);
    }
#undef GO
}
