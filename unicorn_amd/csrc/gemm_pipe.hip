// K1c: persistent, epilogue-overlapped variant of the bf16 MFMA GEMM (gemm.hip) for the plain (1x1, no GroupNorm
// statistics) contractions of the path: the ConvNeXt pointwise MLPs, the 2x2/s2 downsamples' successors and the
// transformer Linears (convnext.py Block.forward pwconv1/pwconv2; deformable_transformer.py linear layers).
//
// Why: with one 256x256 block per CU (the accumulators fill the register file) the epilogue of gemm.hip runs with the
// MFMA pipe idle -- measured +40..+240 % on top of the K loop for the MLP shapes (bias + GELU is ~25 VALU issue
// slots per element; the bf16/fp32 output stream is HBM-write bound).  Here a block is PERSISTENT (one per CU,
// tiles walked in XCD-contiguous order), the tile is 256 x 128 with 8 waves (2 per SIMD, 64x64 per wave) and every wave
// keeps TWO accumulator sets: while tile t+1 accumulates, the finished tile t is drained in slices spread over the
// K steps of t+1 (bias/activation -> wave-private LDS transpose -> residual add -> whole-row stores).  The two waves
// that share a SIMD run their slice / MFMA halves of a K step in opposite order, so one wave's VALU + store issue
// sits under the other's MFMAs.  The first operand tile of t+1 is already in flight during the last K step of t.
//
// LDS: 96 KiB operand double buffer (same XOR-swizzled LDS-DMA layout as gemm.hip) + 8 x 8 KiB staging = 160 KiB.
#include "kernels.h"

__device__ u32x4 g_zero_page_pipe = {0u, 0u, 0u, 0u};

#define GLDS16P(gptr, lptr)                                                                            \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),            \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)
#define OPAQUE64P(x)                                     \
    do {                                                 \
        int _lo = (int)(x), _hi = (int)((x) >> 32);      \
        asm volatile("" : "+v"(_lo), "+v"(_hi));         \
        (x) = ((long)_hi << 32) | (unsigned)_lo;         \
    } while (0)

namespace {
constexpr int WM = 4, WN = 2, TM = 2, TN = 2, NW = WM * WN;
constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, BK = 64;       // 256 x 128 x 64
constexpr int CPR = BK / 8, RPP = 64 / CPR;                        // 8 chunks per 128-B LDS row, 8 rows per 1-KiB DMA piece
constexpr int A_PC = BM / RPP / NW, B_PC = BN / RPP / NW;          // 4 + 2 pieces per wave per K step
constexpr int CW = 32 * TN;                                        // wave tile width (channels)
constexpr int LDS_OPER = 2 * (BM + BN) * BK * 2;                   // 98304
constexpr int LDS_STAGE = NW * 32 * CW * 4;                        // 65536
constexpr int NSLICE = 2 * TM;                                     // (phase A, phase B) per 32-row slab

struct Tile { int m0, n0; };
__device__ __forceinline__ Tile tile_of(int L, int nbm, int nbn) {
    // N is cut into chunks of 8 tiles; inside a chunk tiles run M-major (see gemm.hip)
    constexpr int GN = 8;
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

template <int ACT>
__device__ __forceinline__ float act_ct(float x) {
    return act_fast<ACT>(x);
}

// phase A of slab I: bias + activation in the MFMA layout, float4 chunks to the wave's LDS tile [row][chunk ^ (row & 7)]
template <int I, int ACT>
__device__ __forceinline__ void drain_a(const float* __restrict__ bias, int N, const f32x16 (&acc)[TM][TN], float* st, int nw0,
                                        int lane) {
    const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = nw0 + j * 32 + 8 * g + 4 * fh;
            const int cb = col < N ? col : 0;                        // columns past N are never stored
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + cb);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_ct<ACT>(acc[I][j][4 * g + e] + b4[e]);
            const int c = j * 8 + 2 * g + fh;
            *reinterpret_cast<f32x4*>(st + fr * CW + ((c ^ (fr & 7)) << 2)) = v;
        }
    }
    wave_fence();
}
// phase B: whole rows out (+ residual), rows rbase .. rbase+31 of the output
template <bool OUTF>
__device__ __forceinline__ void drain_b(const GemmArgs& p, const float* st, int rbase, int nw0, int lane) {
    if (!OUTF) {
        constexpr int Q = CW / 8, RPI = 64 / Q;
        const int q = lane % Q, rr = lane / Q;
        const int col = nw0 + 8 * q;
#pragma unroll
        for (int t = 0; t < 32 / RPI; ++t) {
            const int r = t * RPI + rr, row = rbase + r;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(st + r * CW + (((2 * q) ^ (r & 7)) << 2));
            const f32x4 hi = *reinterpret_cast<const f32x4*>(st + r * CW + (((2 * q + 1) ^ (r & 7)) << 2));
            if (row < p.M && col < p.N) {
                bf16x2 w0 = {(bf16)lo[0], (bf16)lo[1]}, w1 = {(bf16)lo[2], (bf16)lo[3]};
                bf16x2 w2 = {(bf16)hi[0], (bf16)hi[1]}, w3 = {(bf16)hi[2], (bf16)hi[3]};
                u32x4 o4 = {__builtin_bit_cast(unsigned, w0), __builtin_bit_cast(unsigned, w1),
                            __builtin_bit_cast(unsigned, w2), __builtin_bit_cast(unsigned, w3)};
                *reinterpret_cast<u32x4*>(p.outB + (size_t)row * p.ldb + col) = o4;
            }
        }
    } else {
        constexpr int CPRo = CW / 4, RPI = 64 / CPRo;
        const int c = lane % CPRo, rr = lane / CPRo;
        const int col = nw0 + 4 * c;
        const bool has_res = p.res != nullptr, has_b = p.outB != nullptr;
#pragma unroll
        for (int t = 0; t < 32 / RPI; ++t) {
            const int r = t * RPI + rr, row = rbase + r;
            f32x4 v = *reinterpret_cast<const f32x4*>(st + r * CW + ((c ^ (r & 7)) << 2));
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
    }
    wave_fence();
}
}  // namespace

// requirements (checked by launch_gemm_pipe): K % 64 == 0, bias != null, act in {none, relu, gelu} on every column, no outF
// row remap, vector-aligned operands (GemmArgs::epi)
template <int ACT, bool OUTF>
__global__ __launch_bounds__(64 * NW) void gemm_bf16_pipe_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* As = reinterpret_cast<bf16*>(smem);                       // [2][BM*BK]
    bf16* Bs = As + 2 * BM * BK;                                    // [2][BN*BK]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    float* st = reinterpret_cast<float*>(smem + LDS_OPER) + wave * (32 * CW);

    // ---- this block's tile list: XCD x (= blockIdx % 8) owns a contiguous range of the tile order, its blocks
    // (slots) take consecutive tiles round-robin so the tiles in flight on one XCD share operand panels in its L2 ----
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

    // ---- per-lane DMA sources (swizzle on the source side, see gemm.hip): byte offsets of this lane's 16-B chunk ----
    const int lrow = lane / CPR;
    const int lch = (lane % CPR) ^ (((RPP * wave + lrow) >> 1) & (CPR - 1));
    const char* abase = reinterpret_cast<const char*>(p.A) + lch * 16;
    const char* wbase = reinterpret_cast<const char*>(p.W) + lch * 16;
    const int rsel = RPP * wave + lrow;                 // row of piece 0 inside a tile; piece i adds RPP*NW*i
    long aoff[A_PC], woff[B_PC];                        // current tile
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
        for (int i = 0; i < A_PC; ++i) GLDS16P(abase + aoff[i] + kt * (BK * 2), adst + i * NW * 1024);
#pragma unroll
        for (int i = 0; i < B_PC; ++i) GLDS16P(wbase + woff[i] + kt * (BK * 2), bdst + i * NW * 1024);
    };

    f32x16 acc[TM][TN], prev[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; prev[i][j][r] = 0.f; }

    const int nk = p.Kpad / BK;
    const int spk = (NSLICE + nk - 1) / nk;         // drain slices per slot
    const int fr = lane & 31, fh = lane >> 5;
    // Flat loop over half-steps: an "M" half-step is one K step (barrier, prefetch, 16 MFMAs), a "D" half-step drains up
    // to spk slices of the previous tile.  Waves w and w+4 share a SIMD and run the two kinds in opposite order
    // (D M D M ... vs M D M D ...), so one wave's VALU / LDS / store issue sits under the other's MFMAs; each kind exists
    // once in the code.
    const int m_par = ((wave >> 2) & 1) && !(p.dbg & 1) ? 0 : 1;   // parity of this wave's M half-steps
    const int gsteps = count * nk;                  // K steps over all tiles of this block
    const int nhalf = 2 * (gsteps + (NSLICE + spk - 1) / spk) + 2;
    int pending = NSLICE;                           // next drain slice of `prev` (NSLICE = nothing to drain)
    int pm0 = 0, pn0 = 0;                           // origin of the tile being drained
    int buf = 0, kt = 0, t = 0, g = 0;
    Tile cur = tile_of(first, nbm, nbn);
    set_tile(cur);
    issue(0, 0);
    // fragment read offsets (bytes) inside an operand buffer; K sub-step kk flips chunk bits via XOR
    const int frow_a = wm * 32 * TM + fr, frow_b = wn * 32 * TN + fr;
#pragma unroll 1
    for (int h = 0; h < nhalf; ++h) {
        if ((h & 1) == m_par) {
            if (g >= gsteps) continue;
            __syncthreads();                          // K tile landed; every wave is done with the other buffer
            if (kt + 1 < nk) {
                issue(kt + 1, buf ^ 1);
            } else if (t + 1 < count) {               // first K tile of the next output tile
                cur = tile_of(first + (t + 1) * stride, nbm, nbn);
                set_tile(cur);
                issue(0, buf ^ 1);
            }
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
            // pin the interleave: the 4 fragment reads of sub-step kk+1 issue BEFORE the 4 MFMAs of sub-step kk
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                if (kk + 1 < BK / 16) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
            buf ^= 1;
            ++g;
            if (++kt == nk) {                         // tile finished: hand the accumulators to the drain
                kt = 0;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        prev[i][j] = acc[i][j];
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                    }
                const Tile done = tile_of(first + t * stride, nbm, nbn);
                pm0 = done.m0; pn0 = done.n0;
                pending = (p.dbg & 2) ? NSLICE : 0;     // ablation: no drain at all
                ++t;
            }
        } else {
            const int nw0 = pn0 + wn * CW;
            const int rb = pm0 + wm * 32 * TM;
#pragma unroll 1
            for (int u = 0; u < spk && pending < NSLICE; ++u, ++pending) {
                if (pending == 0) drain_a<0, ACT>(p.bias, p.N, prev, st, nw0, lane);
                else if (pending == 2) drain_a<1, ACT>(p.bias, p.N, prev, st, nw0, lane);
                else drain_b<OUTF>(p, st, rb + (pending >> 1) * 32, nw0, lane);
            }
        }
    }
}

template <int ACT, bool OUTF>
static int launch_pipe_inst(const GemmArgs& a, int grid, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_pipe_kernel<ACT, OUTF>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_OPER + LDS_STAGE) != hipSuccess) {
            uni_set_error("gemm_pipe: cannot reserve %d bytes of LDS", LDS_OPER + LDS_STAGE);
            return -1;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL((gemm_bf16_pipe_kernel<ACT, OUTF>), dim3(grid), dim3(64 * NW), LDS_OPER + LDS_STAGE, s, a);
    return 0;
}

bool gemm_pipe_supported(const GemmArgs& a) {
    const bool conv = a.KH != 1 || a.KW != 1 || a.stride != 1 || a.pad != 0;
    return !conv && !a.stats && a.b32 == FMT_BF16 && a.epi && a.K == a.Kpad && a.bias && a.act_col0 == 0 && a.out_hw == 0 &&
           (a.act == ACT_NONE || a.act == ACT_RELU || a.act == ACT_GELU) && (a.outF || a.outB) && (a.outF || !a.res);
}

int launch_gemm_pipe(const GemmArgs& a, hipStream_t s) {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
        ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        ncu -= ncu % 8;
    }
    const int ntiles = cdiv(a.M, BM) * cdiv(a.N, BN);
    const int grid = (ntiles < ncu || (a.dbg & 4)) ? (ntiles + 7) / 8 * 8 : ((a.dbg & 8) ? 2 * ncu : ncu);   // dbg 4: one tile per block
    const bool f = a.outF != nullptr;
    switch (a.act) {
        case ACT_GELU: return f ? launch_pipe_inst<ACT_GELU, true>(a, grid, s) : launch_pipe_inst<ACT_GELU, false>(a, grid, s);
        case ACT_RELU: return f ? launch_pipe_inst<ACT_RELU, true>(a, grid, s) : launch_pipe_inst<ACT_RELU, false>(a, grid, s);
        default: return f ? launch_pipe_inst<ACT_NONE, true>(a, grid, s) : launch_pipe_inst<ACT_NONE, false>(a, grid, s);
    }
}
