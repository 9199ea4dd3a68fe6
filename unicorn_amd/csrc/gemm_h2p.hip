// PERSISTENT 256 x 256 variant of the split-f16 ("f16x2") GEMM (gemm_h2.hip) for the plain contractions (1x1, no GroupNorm
// statistics): ConvNeXt pointwise MLPs and transformer Linears (convnext.py:41-54 pwconv1 / pwconv2;
// deformable_transformer.py:122-131) -- 60 % of the f16x2 frame.
//
// Same idea as gemm_p44.hip (bf16): one 16-wave block per CU walks its tiles (XCD-contiguous ranges);
//   * the first K slice of tile t+1 is requested BEFORE the epilogue of tile t (into the operand buffer the last K step
//     did not use), so the cold DMA round trip of a tile hides under the previous tile's epilogue;
//   * the epilogue transposes through the just-consumed operand buffer (4 KiB per wave: fp32 [32 rows][32 columns],
//     float4 chunks XOR-swizzled with row & 7) and only ISSUES its stores: they drain under the next tile's K loop
//     (786 MB of split-f16 hidden activations per stage-2 pwconv1 launch = 0.16 ms of pure HBM write time otherwise exposed);
//   * the activation is a compile-time constant and applies to every column (no per-element column-window select).
// K loop = gemm_h2.hip's: 32 k per step (128-byte LDS rows: [8 hi][8 lo] groups), hi.hi + hi.lo + lo.hi on
// v_mfma_f32_32x32x16_f16, fp32 accumulate, source-side swizzle chunk ^ ((row >> 1) & 7).
#include "kernels.h"

#define GLDS16R(gptr, lptr)                                                                            \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),            \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

namespace {
constexpr int WM = 4, WN = 4, TM = 2, TN = 2, NW = WM * WN;
constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, BKE = 32;      // 256 x 256 x 32 k
constexpr int ROWB = 128, RPP = 8;                                  // LDS row bytes, rows per 1-KiB DMA piece
constexpr int A_PC = BM / RPP / NW, B_PC = BN / RPP / NW;           // 2 + 2 pieces per wave per K step
constexpr int LDS_BYTES = 2 * (BM + BN) * ROWB;                     // 131072

struct Tile { int m0, n0; };
__device__ __forceinline__ Tile tile_of(int L, int nbm, int nbn) {
    constexpr int GN = 8;           // N is cut into chunks of 8 tiles; inside a chunk tiles run M-major (see gemm.hip)
    const int per_chunk = nbm * GN;
    const int c = L / per_chunk;
    const int wc = min(GN, nbn - c * GN);
    const int rem = L - c * per_chunk;
    const int bm = rem / wc;
    return {bm * BM, (c * GN + rem - bm * wc) * BN};
}
__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
}  // namespace

// requirements (gemm_h2p_supported): plain GEMM, K % 32 == 0, bias != null, act in {none, relu, gelu} on every column,
// no outF row remap, vector-aligned operands (GemmArgs::epi)
template <int ACT, bool OUTF>
__global__ __launch_bounds__(64 * NW) void gemm_h2p_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                                                // [2][BM][128 B]
    char* Bs = smem + 2 * BM * ROWB;                                // [2][BN][128 B]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 31, fh = lane >> 5;

    const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
    int first, stride, count;
    {
        const int ntiles = nbm * nbn;
        const int x = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
        const int q = ntiles >> 3, r = ntiles & 7;
        const int start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
        const int cnt = q + (x < r ? 1 : 0);
        first = start + slot;
        stride = nslots;
        count = slot < cnt ? (cnt - slot + nslots - 1) / nslots : 0;
    }
    if (count == 0) return;

    // ---- per-lane DMA sources (swizzle on the source side): lane -> row lane/8 of its piece, logical chunk lch
    // (k group lch >> 1, half lch & 1: 0 = hi, 1 = lo); the 16-byte piece of element group g of a row sits at row + 32 g + 16 half
    const int lrow = lane >> 3;
    const int lch = (lane & 7) ^ (((RPP * wave + lrow) >> 1) & 7);
    const char* abase = reinterpret_cast<const char*>(p.A) + lch * 16;
    const char* wbase = reinterpret_cast<const char*>(p.W) + lch * 16;
    const int rsel = RPP * wave + lrow;                 // row of piece 0 inside a tile; piece i adds RPP*NW*i
    long aoff[A_PC], woff[B_PC];
    auto set_tile = [&](Tile tl) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < A_PC; ++i) {
            int m = tl.m0 + rsel + RPP * NW * i;
            m = m < p.M ? m : p.M - 1;
            aoff[i] = (long)m * p.lda * 4;
        }
#pragma unroll
        for (int i = 0; i < B_PC; ++i) woff[i] = (long)(tl.n0 + rsel + RPP * NW * i) * p.Kpad * 4;
    };
    auto issue = [&](int kt, int buf) __attribute__((always_inline)) {
        char* adst = As + buf * BM * ROWB + wave * 1024;
        char* bdst = Bs + buf * BN * ROWB + wave * 1024;
#pragma unroll
        for (int i = 0; i < A_PC; ++i) GLDS16R(abase + aoff[i] + kt * ROWB, adst + i * NW * 1024);
#pragma unroll
        for (int i = 0; i < B_PC; ++i) GLDS16R(wbase + woff[i] + kt * ROWB, bdst + i * NW * 1024);
    };

    f32x16 acc[TM][TN];
    const int nk = p.K / BKE;
    const int frow_a = wm * 32 * TM + fr, frow_b = wn * 32 * TN + fr;
    int buf = 0;
    Tile cur = tile_of(first, nbm, nbn);
    set_tile(cur);
    issue(0, 0);
#pragma unroll 1
    for (int t = 0; t < count; ++t) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll 1
        for (int kt = 0; kt < nk; ++kt) {
            __syncthreads();                          // K slice landed; every wave is done with the other buffer (and its staging)
            if (kt + 1 < nk) issue(kt + 1, buf ^ 1);
            const char* a = As + buf * BM * ROWB;
            const char* b = Bs + buf * BN * ROWB;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int ch = 2 * (2 * kk + fh);
                f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = frow_a + i * 32, sw = (row >> 1) & 7;
                    ah[i] = *reinterpret_cast<const f16x8*>(a + row * ROWB + ((ch ^ sw) << 4));
                    al[i] = *reinterpret_cast<const f16x8*>(a + row * ROWB + (((ch + 1) ^ sw) << 4));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int row = frow_b + j * 32, sw = (row >> 1) & 7;
                    bh[j] = *reinterpret_cast<const f16x8*>(b + row * ROWB + ((ch ^ sw) << 4));
                    bl[j] = *reinterpret_cast<const f16x8*>(b + row * ROWB + (((ch + 1) ^ sw) << 4));
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al[i], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
            }
            buf ^= 1;
        }
        // ---- tile done.  `buf` = buffer of the next K step (free: last read one step ago), buf^1 = just consumed.
        __syncthreads();                              // every wave finished reading the last K slice
        const Tile done = cur;
        if (t + 1 < count) {                          // request the next tile's first K slice before draining this one
            cur = tile_of(first + (t + 1) * stride, nbm, nbn);
            set_tile(cur);
            issue(0, buf);
        }
        // staging tile of this wave inside the consumed operand buffer: waves 0-7 in the A half, 8-15 in the B half
        char* st = (wave < 8 ? As + (buf ^ 1) * BM * ROWB : Bs + (buf ^ 1) * BN * ROWB) + (wave & 7) * 4096;
        const int nw0 = done.n0 + wn * 32 * TN;
        const int rb0 = done.m0 + wm * 32 * TM;
        if (p.dbg & 16) {                             // ablation: no drain, accumulators kept live
            float sacc = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
            if (sacc == 1.2345e-30f && p.outF) p.outF[0] = sacc;
            continue;
        }
        const float ws = p.wscale;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float rs_mean = 0.f, rs_rstd = 1.f;       // LayerNorm of the A row folded in (GemmArgs::rowstat): rstd * (acc - mean * colsum) + bias
            if (p.rowstat) {
                const int row = rb0 + i * 32 + fr;
                const float2 st2 = *reinterpret_cast<const float2*>(p.rowstat + 2 * (size_t)(row < p.M ? row : p.M - 1));
                rs_mean = st2.x; rs_rstd = st2.y;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                // fp32 [32 rows][32 cols] of block (i, j): 128-B rows, float4 chunks XOR-swizzled with row & 7
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = nw0 + j * 32 + 8 * g + 4 * fh;
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + (col < p.N ? col : 0));
                    f32x4 v;
                    if (p.rowstat) {
                        const f32x4 cs = *reinterpret_cast<const f32x4*>(p.colsum + (col < p.N ? col : 0));
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = act_fast<ACT>(fmaf(rs_rstd, fmaf(-rs_mean, cs[e], acc[i][j][4 * g + e] * ws), b4[e]));
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = act_fast<ACT>(fmaf(acc[i][j][4 * g + e], ws, b4[e]));
                    }
                    *reinterpret_cast<f32x4*>(st + fr * 128 + (((2 * g + fh) ^ (fr & 7)) << 4)) = v;
                }
                wave_fence();
                if (OUTF) {        // fp32 (+ residual) output, optional operand-format copy: 4 channels per lane, 8 rows per instruction
                    const bool has_res = p.res != nullptr, has_b = p.outB != nullptr;
                    const int c = lane & 7, rr = lane >> 3;
                    const int col = nw0 + j * 32 + 4 * c;
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        const int r = tt * 8 + rr, row = rb0 + i * 32 + r;
                        f32x4 v = *reinterpret_cast<const f32x4*>(st + r * 128 + ((c ^ (r & 7)) << 4));
                        if (row < p.M && col < p.N) {
                            if (has_res) v += *reinterpret_cast<const f32x4*>(p.res + (size_t)row * p.ldr + col);
                            *reinterpret_cast<f32x4*>(p.outF + (size_t)row * p.ldf + col) = v;
                            if (has_b) act_store4(p.outB, (size_t)row * p.ldb + col, v[0], v[1], v[2], v[3], FMT_H2);
                        }
                    }
                } else {           // operand-format output only: 8 channels (one 32-byte [hi | lo] group) per lane, 16 rows per instruction
                    const int c = lane & 3, rr = lane >> 2;
                    const int col = nw0 + j * 32 + 8 * c;
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const int r = tt * 16 + rr, row = rb0 + i * 32 + r;
                        const f32x4 lo = *reinterpret_cast<const f32x4*>(st + r * 128 + (((2 * c) ^ (r & 7)) << 4));
                        const f32x4 hi = *reinterpret_cast<const f32x4*>(st + r * 128 + (((2 * c + 1) ^ (r & 7)) << 4));
                        if (row < p.M && col < p.N) {
                            const float v8[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                            act_store8(p.outB, (size_t)row * p.ldb + col, v8, FMT_H2);
                        }
                    }
                }
                wave_fence();
            }
        }
    }
}

template <int ACT, bool OUTF>
static int launch_h2p_inst(const GemmArgs& a, int grid, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h2p_kernel<ACT, OUTF>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) {
            uni_set_error("gemm_h2p: cannot reserve %d bytes of LDS", LDS_BYTES);
            return -1;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL((gemm_h2p_kernel<ACT, OUTF>), dim3(grid), dim3(64 * NW), LDS_BYTES, s, a);
    return 0;
}

bool gemm_h2p_supported(const GemmArgs& a) {
    const bool conv = a.KH != 1 || a.KW != 1 || a.stride != 1 || a.pad != 0;
    return !conv && !a.stats && a.b32 == FMT_H2 && a.epi && a.K % BKE == 0 && a.bias && a.act_col0 == 0 && a.out_hw == 0 &&
           (a.act == ACT_NONE || a.act == ACT_RELU || a.act == ACT_GELU) && (a.outF || a.outB) && (a.outF || !a.res);
}

int launch_gemm_h2p(const GemmArgs& a, hipStream_t s) {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
        ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        ncu -= ncu % 8;
    }
    const int ntiles = cdiv(a.M, BM) * cdiv(a.N, BN);
    const int grid = ntiles < ncu ? (ntiles + 7) / 8 * 8 : ncu;
    const bool f = a.outF != nullptr;
    switch (a.act) {
        case ACT_GELU: return f ? launch_h2p_inst<ACT_GELU, true>(a, grid, s) : launch_h2p_inst<ACT_GELU, false>(a, grid, s);
        case ACT_RELU: return f ? launch_h2p_inst<ACT_RELU, true>(a, grid, s) : launch_h2p_inst<ACT_RELU, false>(a, grid, s);
        default: return f ? launch_h2p_inst<ACT_NONE, true>(a, grid, s) : launch_h2p_inst<ACT_NONE, false>(a, grid, s);
    }
}
