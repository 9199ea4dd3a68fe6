// K1d: PERSISTENT 256 x 256 variant of the bf16 MFMA GEMM (same 16-wave tile, LDS-DMA double buffer and swizzle as cfg 44 of
// gemm.hip) for the plain contractions (1x1, no GroupNorm statistics): ConvNeXt pointwise MLPs and transformer Linears
// (convnext.py:41-54 pwconv1 / pwconv2; deformable_transformer.py:122-131).
//
// What it removes (tools/gemm_epi.py ablations): with one 1024-thread block per CU every output tile pays a
// block launch, a cold prologue (first DMA round trip) and the drain of its stores before the next block may start --
// ~7 us per tile next to 19-26 us of K loop.  Here one block per CU walks its tiles:
//   * the first operand K-slice of tile t+1 is requested BEFORE the epilogue of tile t (into the buffer the last K step did
//     not use), so it lands while the epilogue runs;
//   * the epilogue transposes through the OTHER operand buffer (4 KiB per wave: bf16 [32][64] for bf16-only outputs, fp32
//     [32][32] otherwise) and only issues its stores; they drain under the next tile's K loop;
//   * tile order: XCD-contiguous ranges, blocks of one XCD take consecutive tiles (operand panels shared in its L2).
// The K loop itself is unchanged (one barrier per step).
#include "kernels.h"

__device__ u32x4 g_zero_page_p44 = {0u, 0u, 0u, 0u};

#define GLDS16Q(gptr, lptr)                                                                            \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),            \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

namespace {
constexpr int WM = 4, WN = 4, TM = 2, TN = 2, NW = WM * WN;
constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, BK = 64;       // 256 x 256 x 64
constexpr int CPR = BK / 8, RPP = 64 / CPR;                        // 8 chunks per 128-B LDS row, 8 rows per 1-KiB DMA piece
constexpr int A_PC = BM / RPP / NW, B_PC = BN / RPP / NW;          // 2 + 2 pieces per wave per K step
constexpr int LDS_BYTES = 2 * (BM + BN) * BK * 2;                  // 131072

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

// requirements (gemm_p44_supported): K % 64 == 0, bias != null, act in {none, relu, gelu} on every column, no outF row remap,
// vector-aligned operands (GemmArgs::epi)
template <int ACT, bool OUTF>
__global__ __launch_bounds__(64 * NW) void gemm_bf16_p44_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* As = reinterpret_cast<bf16*>(smem);                       // [2][BM*BK]
    bf16* Bs = As + 2 * BM * BK;                                    // [2][BN*BK]
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

    // ---- per-lane DMA sources (swizzle on the source side, see gemm.hip) ----
    const int lrow = lane / CPR;
    const int lch = (lane % CPR) ^ (((RPP * wave + lrow) >> 1) & (CPR - 1));
    const char* abase = reinterpret_cast<const char*>(p.A) + lch * 16;
    const char* wbase = reinterpret_cast<const char*>(p.W) + lch * 16;
    const int rsel = RPP * wave + lrow;                 // row of piece 0 inside a tile; piece i adds RPP*NW*i
    long aoff[A_PC], woff[B_PC];
    auto set_tile = [&](Tile tl) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < A_PC; ++i) {
            int m = tl.m0 + rsel + RPP * NW * i;
            m = m < p.M ? m : p.M - 1;
            aoff[i] = (long)m * p.lda * 2;
        }
#pragma unroll
        for (int i = 0; i < B_PC; ++i) woff[i] = (long)(tl.n0 + rsel + RPP * NW * i) * p.Kpad * 2;
    };
    auto issue = [&](int kt, int buf) __attribute__((always_inline)) {
        char* adst = reinterpret_cast<char*>(As + buf * BM * BK) + wave * 1024;
        char* bdst = reinterpret_cast<char*>(Bs + buf * BN * BK) + wave * 1024;
#pragma unroll
        for (int i = 0; i < A_PC; ++i) GLDS16Q(abase + aoff[i] + kt * (BK * 2), adst + i * NW * 1024);
#pragma unroll
        for (int i = 0; i < B_PC; ++i) GLDS16Q(wbase + woff[i] + kt * (BK * 2), bdst + i * NW * 1024);
    };

    f32x16 acc[TM][TN];
    const int nk = p.Kpad / BK;
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
            const bf16* a = As + buf * BM * BK;
            const bf16* b = Bs + buf * BN * BK;
            bf16x8 fa[2][TM], fb[2][TN];
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                if (kk == 0) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int row = frow_a + i * 32;
                        fa[0][i] = *reinterpret_cast<const bf16x8*>(a + row * BK + ((fh ^ ((row >> 1) & (CPR - 1))) << 3));
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int row = frow_b + j * 32;
                        fb[0][j] = *reinterpret_cast<const bf16x8*>(b + row * BK + ((fh ^ ((row >> 1) & (CPR - 1))) << 3));
                    }
                }
                if (kk + 1 < BK / 16) {
                    const int ch = (kk + 1) * 2 + fh, slot = (kk + 1) & 1;
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int row = frow_a + i * 32;
                        fa[slot][i] = *reinterpret_cast<const bf16x8*>(a + row * BK + ((ch ^ ((row >> 1) & (CPR - 1))) << 3));
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int row = frow_b + j * 32;
                        fb[slot][j] = *reinterpret_cast<const bf16x8*>(b + row * BK + ((ch ^ ((row >> 1) & (CPR - 1))) << 3));
                    }
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)   // swapped operands: lane -> pixel row, 4 consecutive channels per quad
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kk & 1][j], fa[kk & 1][i], acc[i][j], 0, 0, 0);
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
        char* st = reinterpret_cast<char*>(wave < 8 ? (As + (buf ^ 1) * BM * BK) : (Bs + (buf ^ 1) * BN * BK)) + (wave & 7) * 4096;
        const int nw0 = done.n0 + wn * 32 * TN;
        const int rb0 = done.m0 + wm * 32 * TM;
        if (p.dbg & 16) {                             // ablation (tools/gemm_epi.py): no drain, accumulators kept live
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
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (!OUTF) {
                // bf16 [32 rows][64 cols] (128-B rows); 16-B chunks XOR-swizzled with (row >> 1) & 7
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int col = nw0 + j * 32 + 8 * g + 4 * fh;
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + (col < p.N ? col : 0));
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = act_fast<ACT>(acc[i][j][4 * g + e] + b4[e]);
                        bf16x2 w0 = {(bf16)v[0], (bf16)v[1]}, w1 = {(bf16)v[2], (bf16)v[3]};
                        u32x2 o2 = {__builtin_bit_cast(unsigned, w0), __builtin_bit_cast(unsigned, w1)};
                        const int chunk = (j * 4 + g) ^ ((fr >> 1) & 7);             // 16-B chunk holding cols 8g..8g+7 of block j
                        *reinterpret_cast<u32x2*>(st + fr * 128 + chunk * 16 + fh * 8) = o2;
                    }
                }
                wave_fence();
                const int q = lane & 7, rr = lane >> 3;
                const int col = nw0 + 8 * q;
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const int r = tt * 8 + rr, row = rb0 + i * 32 + r;
                    const u32x4 o4 = *reinterpret_cast<const u32x4*>(st + r * 128 + ((q ^ ((r >> 1) & 7)) << 4));
                    if (row < p.M && col < p.N) *reinterpret_cast<u32x4*>(p.outB + (size_t)row * p.ldb + col) = o4;
                }
                wave_fence();
            } else {
                // fp32 [32 rows][32 cols] per 32-column block j (128-B rows); float4 chunks XOR-swizzled with row & 7
                const bool has_res = p.res != nullptr, has_b = p.outB != nullptr;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int col = nw0 + j * 32 + 8 * g + 4 * fh;
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + (col < p.N ? col : 0));
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = act_fast<ACT>(acc[i][j][4 * g + e] + b4[e]);
                        *reinterpret_cast<f32x4*>(st + fr * 128 + (((2 * g + fh) ^ (fr & 7)) << 4)) = v;
                    }
                    wave_fence();
                    const int c = lane & 7, rr = lane >> 3;
                    const int col = nw0 + j * 32 + 4 * c;
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        const int r = tt * 8 + rr, row = rb0 + i * 32 + r;
                        f32x4 v = *reinterpret_cast<const f32x4*>(st + r * 128 + ((c ^ (r & 7)) << 4));
                        if (row < p.M && col < p.N) {
                            if (has_res) v += *reinterpret_cast<const f32x4*>(p.res + (size_t)row * p.ldr + col);
                            *reinterpret_cast<f32x4*>(p.outF + (size_t)row * p.ldf + col) = v;
                            if (has_b) {
                                bf16x2 w0 = {(bf16)v[0], (bf16)v[1]}, w1 = {(bf16)v[2], (bf16)v[3]};
                                u32x2 o2 = {__builtin_bit_cast(unsigned, w0), __builtin_bit_cast(unsigned, w1)};
                                *reinterpret_cast<u32x2*>(p.outB + (size_t)row * p.ldb + col) = o2;
                            }
                        }
                    }
                    wave_fence();
                }
            }
        }
    }
}

template <int ACT, bool OUTF>
static int launch_p44_inst(const GemmArgs& a, int grid, hipStream_t s) {
    static DevOnce attr_once;
    UNI_LDS_OPTIN(attr_once, "gemm_p44", LDS_BYTES, reinterpret_cast<const void*>(&gemm_bf16_p44_kernel<ACT, OUTF>));
    hipLaunchKernelGGL((gemm_bf16_p44_kernel<ACT, OUTF>), dim3(grid), dim3(64 * NW), LDS_BYTES, s, a);
    return 0;
}

bool gemm_p44_supported(const GemmArgs& a) {      // plain GEMM, K % 64 == 0, bias, none/relu/gelu on every column, aligned operands
    const bool conv = a.KH != 1 || a.KW != 1 || a.stride != 1 || a.pad != 0;
    return !conv && !a.stats && a.b32 == FMT_BF16 && a.epi && a.K == a.Kpad && a.bias && a.act_col0 == 0 && a.out_hw == 0 &&
           (a.act == ACT_NONE || a.act == ACT_RELU || a.act == ACT_GELU) && (a.outF || a.outB) && (a.outF || !a.res);
}

int launch_gemm_p44(const GemmArgs& a, hipStream_t s) {
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
        case ACT_GELU: return f ? launch_p44_inst<ACT_GELU, true>(a, grid, s) : launch_p44_inst<ACT_GELU, false>(a, grid, s);
        case ACT_RELU: return f ? launch_p44_inst<ACT_RELU, true>(a, grid, s) : launch_p44_inst<ACT_RELU, false>(a, grid, s);
        default: return f ? launch_p44_inst<ACT_NONE, true>(a, grid, s) : launch_p44_inst<ACT_NONE, false>(a, grid, s);
    }
}
