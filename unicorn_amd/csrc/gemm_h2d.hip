// K1b / K2 / K5, f16x2 operands: the DEEP-PIPELINE variant of gemm_h2_kernel for launches that put at most ~one block on a CU
// (single-frame calls: M = 1000 .. 16000 rows, see DESIGN.md "single frame").
//
// gemm_h2_kernel requests K slice kt+1 when slice kt starts and waits for it with vmcnt(0) + barrier one slice later: a 128 x 128
// slice is 24 MFMAs = ~770 clk per wave, an LDS-DMA request under load takes 2000+ clk to land, so with ONE resident block per CU
// (nothing else to switch to) every K step waits for memory and those launches sat at 220-230 TF-eq (27 % of the MFMA peak)
// while the same shapes reach 350+ at 16 frames, where 2-3 blocks share a CU.  Here a block keeps NST - 1 slices in flight in an
// NST-deep LDS ring: requests are counted (`s_waitcnt vmcnt((NST-2) * PER)`: only the OLDEST slice has to have landed), the
// barrier is a raw `s_barrier` (no vmcnt(0) drain), and the fragments of the second 16-k half are read under the MFMAs of the
// first.  Tiles: 4 waves, 128 x 128 / 128 x 96 (one block per CU, 4 slices deep) and 64 x 128 / 64 x 64 (3 deep, 2 / 3 blocks per CU)
// -- 128 x 96 exists because N = 768 over 4000 rows is exactly 256 tiles of it (192 of 128 x 128: a quarter of the chip idle) -- and
// 8 waves, 256 x 192 with 2 slots (the 256-wide problems whose N = 192 k would pad the ping-pong kernel's 256 x 256 tiles).
// Same operand format, loader geometry, swizzle, epilogue (gemm_epi.h) and split-K protocol (slab + splitk_reduce_kernel) as
// gemm_h2.hip; results are bit-identical to gemm_h2_kernel's for the same K order (same MFMA sequence per accumulator).
#include "kernels.h"
#include <cstdlib>

#define D_LDS(addr) (*reinterpret_cast<const __attribute__((address_space(3))) f16x8*>((size_t)(addr)))

#include "gemm_epi.h"

namespace {
__device__ __forceinline__ void d_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
template <int N>
__device__ __forceinline__ void d_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
}  // namespace

// SPEC (round 5, 4-wave tiles only): the block carries 4 MORE waves that do nothing but issue the LDS-DMA requests -- one producer and one
// consumer per SIMD.  With one wave per SIMD the 7 requests of a K step sit in the same in-order instruction stream as the MFMAs: a request
// that waits for a slot of the (saturated) vector-memory queue stalls the wave's MFMA issue as well.  tools/feed_probe.hip measures that
// skeleton (the 128 x 96 tile's request stream + fragment reads + MFMAs, no epilogue, 4000 x 768 x 3072): 54.3 us with the requests between
// the MFMAs, 47.9 us with producer waves; the request pattern (row-major, padded, K-blocked) and the ring depth (3 .. 8) change nothing, and
// neither does moving one operand to plain VGPR loads (LDS-DMA and VGPR loads share ONE L2 -> CU path: 58 B/clk/CU for either or any mix,
// profiles/r05_feed_probe.txt).  The producers leave after the K loop (through the first barrier of the epilogue); the consumers' MFMA order
// per accumulator is unchanged, so results are bit-identical to the 4-wave tile's.  MEASURED IN THE MODEL: slower (see launch_gemm_h2d), so
// cfg 422 / 423 are opt-in.
template <int WM, int WN, int TM, int TN, bool CONV, bool STATS, int NST, bool SPEC = false>
__global__ __launch_bounds__(64 * WM * WN * (SPEC ? 2 : 1), ((WM * WN > 4 || SPEC) ? 2 : 1)) void gemm_h2d_kernel(GemmArgs p) {
    constexpr int NW = WM * WN;
    static_assert(!SPEC || (NW == 4 && !STATS), "producer waves: 4-wave tiles without GroupNorm statistics");
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int BKE = 32, ROWB = 128, CPR = 8, RPP = 8;
    constexpr int A_PC = BM / RPP / NW, B_PC = BN / RPP / NW, PER = A_PC + B_PC;
    constexpr int STAGE = (BM + BN) * ROWB;
    static_assert(A_PC >= 1 && B_PC >= 1 && BM % (RPP * NW) == 0 && BN % (RPP * NW) == 0, "tile vs wave grid");
    static_assert(NST >= 2 && (NST - 2) * PER <= 63, "ring depth vs the 6-bit vmcnt");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lds0 = (int)(size_t)(__attribute__((address_space(3))) char*)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = SPEC && wave_all >= NW;          // waves NW .. 2 NW - 1: DMA only
    const int wave = SPEC ? (wave_all & (NW - 1)) : wave_all;      // a producer issues the pieces of the consumer with the same index
    const int wm = wave / WN, wn = wave % WN;

    const int nbn = (p.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int L;
    {   // XCD-aware bijective block remap (block b runs on XCD b % 8)
        const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    int bm, bn;
    {   // chunks of 8 N tiles, M-major inside a chunk: the blocks of an XCD share A rows and a window of W in its L2
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

    // ---- DMA (the addressing of gemm_h2q.hip): `buffer_load_dwordx4 ... lds` through one descriptor per operand; wave w fills the
    // 1-KiB pieces w, w + NW, ... (8 rows each); lane -> row lane / 8 of the piece, physical 16-B chunk lane % 8, which holds the
    // row's LOGICAL chunk lch = phys ^ ((row >> 1) & 7) (k group lch >> 1, half lch & 1).  Per-lane offsets are constant, the K step
    // (and the conv tap) travel in the scalar offset; rows past M / N and taps outside the image present an offset beyond
    // num_records and read zeros -- a request costs one instruction (plus 3 VALU for a conv tap), which matters here: with one wave
    // per SIMD every address instruction is a hole in the MFMA stream.
    const int lrow = lane >> 3;
    const int lch = (lane & 7) ^ ((4 * wave + (lrow >> 1)) & 7);          // the pieces of a wave are 8 NW rows apart (a multiple of 16): same swizzle
    static_assert(NW == 4 || NW == 8, "swizzle term assumes 4 or 8 waves");
    const int lda4 = p.lda * 4, ldw4 = p.Kpad * 4;
    char* abase = const_cast<char*>(reinterpret_cast<const char*>(p.A));
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.W)) + (long)n0 * ldw4, 0,
                                                                           min(BN, p.N - n0) * ldw4, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_a;
    static_assert(A_PC <= 4 && B_PC <= 4, "offset tables");
    int voa[4], vob[4];           // fixed size: a template-dependent array bound used by the DMA builtin inside a lambda makes hipcc drop the host stubs
#pragma unroll
    for (int i = 0; i < B_PC; ++i) vob[i] = (RPP * (wave + NW * i) + lrow) * ldw4 + lch * 16;
    if (!CONV) {
        rs_a = __builtin_amdgcn_make_buffer_rsrc(abase + (long)m0 * lda4, 0, min(BM, p.M - m0) * lda4, 0x00020000);
#pragma unroll
        for (int i = 0; i < A_PC; ++i) voa[i] = (RPP * (wave + NW * i) + lrow) * lda4 + lch * 16;
    } else {
        // the descriptor starts (pad, pad) pixels before the tile's first input pixel; per-lane offset = the lane's pixel relative to
        // that (26 bits) | 6 flag bits: bit ky = input row of tap row ky inside the image, bit 3 + kx the same for the column
        const int b0 = m0 / p.Mper, q0 = m0 - b0 * p.Mper, oy0 = q0 / p.Wout, ox0 = q0 - oy0 * p.Wout;
        const int pix0 = (b0 * p.Hin + oy0 * p.stride) * p.Win + ox0 * p.stride;
        rs_a = __builtin_amdgcn_make_buffer_rsrc(abase + ((long)pix0 - (p.pad * p.Win + p.pad)) * lda4, 0, 0x7ffffff0, 0x00020000);
#pragma unroll
        for (int i = 0; i < A_PC; ++i) {
            const int m = m0 + RPP * (wave + NW * i) + lrow;
            const bool ok = m < p.M;
            const int mm = ok ? m : m0;
            const int b = mm / p.Mper, q = mm - b * p.Mper, oy = q / p.Wout, ox = q - oy * p.Wout;
            const int off = ((b * p.Hin + oy * p.stride) * p.Win + ox * p.stride - pix0) * lda4 + lch * 16;
            int bits = 0;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int iy = oy * p.stride - p.pad + t, ix = ox * p.stride - p.pad + t;
                bits |= (t < p.KH && iy >= 0 && iy < p.Hin) ? (1 << t) : 0;
                bits |= (t < p.KW && ix >= 0 && ix < p.Win) ? (8 << t) : 0;
            }
            voa[i] = off | ((ok ? bits : 0) << 26);
        }
    }
    const int cs = CONV ? p.Cin / BKE : 1;           // K steps per tap (Cin % 32 == 0: a K step = 32 channels of ONE tap)
    const int csm = 65536 / cs + 1;                  // kt / cs = (kt * csm) >> 16 for kt < 9 cs <= 2^12 (launcher checks)
    const int kwm = p.KW == 3 ? 11 : p.KW == 2 ? 16 : 32;   // tap / KW = (tap * kwm) >> 5 for tap < 9
#define D_DMA(rs, vo, soff, dst) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst), 16, vo, soff, 0, 0)
    // `live` false (past the end of the K range): the requests still go out, with offsets beyond num_records -- the number of
    // requests in flight stays constant, so the K loop has ONE wait count and no branch around the DMA
    auto issue = [&](int kt, int slot, bool live) __attribute__((always_inline)) {
        char* adst = smem + slot * STAGE + wave * 1024;
        char* bdst = adst + BM * ROWB;
        if (!CONV) {
#pragma unroll
            for (int i = 0; i < A_PC; ++i) D_DMA(rs_a, live ? voa[i] : 0x7fffffff, kt * ROWB, adst + i * NW * 1024);
        } else {
            const int tap = (kt * csm) >> 16, cstep = kt - tap * cs;
            const int ky = (tap * kwm) >> 5, kx = tap - ky * p.KW;
            const int soff = live ? (ky * p.Win + kx) * lda4 + cstep * ROWB : 0;
#pragma unroll
            for (int i = 0; i < A_PC; ++i) {
                const bool ok = live && ((voa[i] >> (26 + ky)) & (voa[i] >> (29 + kx)) & 1) != 0;
                D_DMA(rs_a, ok ? (voa[i] & 0x03ffffff) : 0x7fffffff, soff, adst + i * NW * 1024);
            }
        }
#pragma unroll
        for (int i = 0; i < B_PC; ++i) D_DMA(rs_b, live ? vob[i] : 0x7fffffff, kt * ROWB, bdst + i * NW * 1024);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int kt0 = 0, nk = (p.K + BKE - 1) / BKE;           // the packed weights are zero beyond K (Kpad >= roundup(K, 64))
    if (p.splitk > 1) {                                // split-K: this block owns K steps [kt0, nk) of its tile (empty range: zero partials)
        const int per = (nk + p.splitk - 1) / p.splitk;
        kt0 = blockIdx.y * per;
        nk = min(nk, kt0 + per);
    }
    // fragment addresses inside a stage: row r, logical chunk ch at (ch ^ ((r >> 1) & 7)) * 16
    const int fr = lane & 31, fh = lane >> 5;
    int arow[TM], brow[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) arow[i] = wm * 32 * TM + i * 32 + fr;
#pragma unroll
    for (int j = 0; j < TN; ++j) brow[j] = BM + wn * 32 * TN + j * 32 + fr;

    // Schedule of a K step (one wave per SIMD: nothing else hides latency, so the order is pinned with sched_barrier):
    //   wait(slice kt) | barrier | TM TN MFMAs of (kt-1, half 1) from set 1 | read half 0 of kt -> set 0, DMA requests of slice
    //   kt+NST-1 | 2 TM TN MFMAs (set 1) | TM TN MFMAs of (kt, half 0) from set 0 | read half 1 of kt -> set 1 | 2 TM TN MFMAs (set 0)
    // both fragment reads run under MFMAs that do not depend on them; per accumulator the MFMA order is gemm_h2_kernel's.
    f16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
    auto ldfrag = [&](int sbase, int kk, int set) __attribute__((always_inline)) {
        const int ch = 2 * (2 * kk + fh);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int sw = (arow[i] >> 1) & 7, ra = sbase + arow[i] * ROWB;
            ah[set][i] = D_LDS(ra + ((ch ^ sw) << 4));
            al[set][i] = D_LDS(ra + (((ch + 1) ^ sw) << 4));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int sw = (brow[j] >> 1) & 7, rb = sbase + brow[j] * ROWB;
            bh[set][j] = D_LDS(rb + ((ch ^ sw) << 4));
            bl[set][j] = D_LDS(rb + (((ch + 1) ^ sw) << 4));
        }
    };
    // swapped operands (weights = MFMA A): lane -> pixel row, 4 consecutive channels per accumulator quad
    auto mma = [&](int set, int term) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 0 ? bl[set][j] : bh[set][j], term == 1 ? al[set][i] : ah[set][i], acc[i][j], 0, 0, 0);
    };
#pragma unroll
    for (int i = 0; i < TM; ++i) { ah[1][i] = f16x8{}; al[1][i] = f16x8{}; }     // "half 1 of slice kt0-1": zeros, the loop has no first-step branch
#pragma unroll
    for (int j = 0; j < TN; ++j) { bh[1][j] = f16x8{}; bl[1][j] = f16x8{}; }
    if (!SPEC || producer) {
#pragma unroll
        for (int i = 0; i < NST - 1; ++i) issue(kt0 + i, i, kt0 + i < nk);
    }
    int rd = 0, wr = NST - 1;
    __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0): no scalar load pending at the loop head, so the waits on the fragment reads inside can be counted
    for (int kt = kt0; kt < nk; ++kt) {
        if (!SPEC || producer) d_wait_vm<(NST - 2) * PER>();   // slice kt landed (this wave's share); NST - 2 younger ones stay in flight
        __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0): this wave's reads of slice kt-1 are complete (its slot is requested below)
        d_barrier();                                       // ... everybody's share; everybody is done with the slot of slice kt-1
        const int sbase = lds0 + rd * STAGE;
        rd = rd + 1 == NST ? 0 : rd + 1;
        if (SPEC) {
            if (producer) {
                issue(kt + NST - 1, wr, kt + NST - 1 < nk);
            } else {      // the 4-wave schedule without the requests
                __builtin_amdgcn_sched_barrier(0);
                mma(1, 0);
                __builtin_amdgcn_sched_barrier(0);
                ldfrag(sbase, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                mma(1, 1); mma(1, 2);
                __builtin_amdgcn_sched_barrier(0);
                mma(0, 0);
                __builtin_amdgcn_sched_barrier(0);
                ldfrag(sbase, 1, 1);
                __builtin_amdgcn_sched_barrier(0);
                mma(0, 1); mma(0, 2);
                __builtin_amdgcn_sched_barrier(0);
            }
            wr = wr + 1 == NST ? 0 : wr + 1;
            continue;
        }
        // every batch of fragment reads is followed by >= 2 TM TN independent MFMAs before the first MFMA that needs it (hipcc
        // waits with lgkmcnt(0) there: a batch issued later than the one needed would be waited for as well)
        __builtin_amdgcn_sched_barrier(0);
        mma(1, 0);
        __builtin_amdgcn_sched_barrier(0);
        ldfrag(sbase, 0, 0);
        issue(kt + NST - 1, wr, kt + NST - 1 < nk);
        wr = wr + 1 == NST ? 0 : wr + 1;
        __builtin_amdgcn_sched_barrier(0);
        mma(1, 1); mma(1, 2);
        __builtin_amdgcn_sched_barrier(0);
        mma(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        ldfrag(sbase, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(0, 1); mma(0, 2);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (SPEC && producer) {      // every request (the dead ones past the K range too) has landed; leave through the epilogue's first barrier
        d_wait_vm<0>();
        __syncthreads();
        return;
    }
    mma(1, 0); mma(1, 1); mma(1, 2);
    d_wait_vm<0>();
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

template <int WM, int WN, int TM, int TN, bool CONV, int NST>
static int launch_h2d_spec(const GemmArgs& a, hipStream_t s) {      // producer / consumer build: plain (no statistics, unsplit) launches only
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    const int grid = cdiv(a.M, BM) * cdiv(a.N, BN);
    size_t lds = (size_t)NST * (BM + BN) * 128;
    if (lds < (size_t)WM * WN * 32 * 32 * TN * sizeof(float)) lds = (size_t)WM * WN * 32 * 32 * TN * sizeof(float);
    static DevOnce attr_once;
    UNI_LDS_OPTIN(attr_once, "gemm_h2d (producer waves)", lds, reinterpret_cast<const void*>(&gemm_h2d_kernel<WM, WN, TM, TN, CONV, false, NST, true>));
    hipLaunchKernelGGL((gemm_h2d_kernel<WM, WN, TM, TN, CONV, false, NST, true>), dim3(grid), dim3(128 * WM * WN), lds, s, a);
    return 0;
}

template <int WM, int WN, int TM, int TN, bool CONV, int NST>
static int launch_h2d_cfg(const GemmArgs& a, hipStream_t s) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    const int grid = cdiv(a.M, BM) * cdiv(a.N, BN);
    const int gy = a.splitk > 1 ? a.splitk : 1;
    size_t lds = (size_t)NST * (BM + BN) * 128;
    if (lds < (size_t)WM * WN * 32 * 32 * TN * sizeof(float)) lds = (size_t)WM * WN * 32 * 32 * TN * sizeof(float);   // staged epilogue
    if (lds < (WM * BN * 2 + 128) * sizeof(float)) lds = (WM * BN * 2 + 128) * sizeof(float);
    static DevOnce attr_once;
    UNI_LDS_OPTIN(attr_once, "gemm_h2d", lds, reinterpret_cast<const void*>(&gemm_h2d_kernel<WM, WN, TM, TN, CONV, true, NST>),
                  reinterpret_cast<const void*>(&gemm_h2d_kernel<WM, WN, TM, TN, CONV, false, NST>));
    if (a.stats && gy == 1) hipLaunchKernelGGL((gemm_h2d_kernel<WM, WN, TM, TN, CONV, true, NST>), dim3(grid), dim3(64 * WM * WN), lds, s, a);
    else hipLaunchKernelGGL((gemm_h2d_kernel<WM, WN, TM, TN, CONV, false, NST>), dim3(grid, gy), dim3(64 * WM * WN), lds, s, a);
    if (gy > 1) return launch_splitk_reduce(a, s);
    return 0;
}

bool gemm_h2d_has_cfg(int cfg) { return cfg == 322 || cfg == 323 || cfg == 331 || cfg == 332 || cfg == 346 || cfg == 422 || cfg == 423; }

// what the descriptor addressing covers (everything else stays on gemm_h2_kernel)
bool gemm_h2d_supported(const GemmArgs& a, int cfg) {
    const bool conv = a.KH != 1 || a.KW != 1 || a.stride != 1 || a.pad != 0;
    if (!gemm_h2d_has_cfg(cfg) || a.b32 != FMT_H2 || !a.epi || a.K % 32 != 0 || a.K < 64) return false;
    const long rows = cfg == 346 ? 256 : 128;    // tallest tile of the configuration (A rows / weight rows addressed from one descriptor base)
    if (rows * a.lda * 4 >= (1L << 30) || rows * a.Kpad * 4 >= (1L << 30)) return false;
    if (conv) {
        if (a.Cin % 32 != 0 || a.KH > 3 || a.KW > 3 || a.K != a.KH * a.KW * a.Cin || a.Cin / 32 > 448) return false;
        // the input pixels of one row tile must lie within 2^26 bytes of its first one (the per-lane offsets carry 6 flag bits)
        const long span_pix = (rows / a.Wout + 2) * a.stride * a.Win + rows * a.stride + 3L * a.Win;
        if (span_pix * a.lda * 4 >= (1L << 26)) return false;
    }
    return true;
}

// cfg: 322 = 128 x 128 and 323 = 128 x 96 (4-deep ring, one block per CU), 332 = 64 x 128 (3-deep, two per CU), 331 = 64 x 64 (3-deep, three
// per CU), 346 = 256 x 192 (8 waves = two per SIMD, 2 slots of 56 KiB: the K step of a wave is 36 MFMAs, long enough to cover the DMA of the
// next slice; for 256-wide problems whose N = 192 k pads 256 x 256 tiles -- launch_gemm_h2).  256 x 128 with 3 slots was measured equal to it.  (Measured and dropped: 128 x 64, 5-deep 64 x 64, 3-deep 128 x 128 -- never the best choice, tools/gemm_b1_bench.py; 128 x 192 for the
// stage-2 pwconv1 of one frame -- 512 tiles = 2 full rounds -- loses to the ping-pong kernel at 75 % fill, 10.2 vs 9.8 ms per frame.)
int launch_gemm_h2d(const GemmArgs& a, int cfg, bool conv, hipStream_t s) {
    UNI_REQUIRE(gemm_h2d_supported(a, cfg), "gemm(h2, deep): cfg %d does not cover this problem (K %% 32, Cin %% 32, 3x3 taps at most, staged epilogue)", cfg);
#define GOD(WM, WN, TM, TN, NST) return conv ? launch_h2d_cfg<WM, WN, TM, TN, true, NST>(a, s) : launch_h2d_cfg<WM, WN, TM, TN, false, NST>(a, s)
    // 422 / 423 = 322 / 323 with producer waves (SPEC); launches they do not cover (GroupNorm statistics, K ranges) take the 4-wave build.
    // NOT the launcher's choice: in the model the producer build is SLOWER (stage-2 pwconv2 of one frame 76.9 -> 88.7 us, 9.74 -> 9.90 ms per
    // frame, same box) although the skeleton probe gains 12 % -- with real (random) operands, the epilogue and 8 waves at every barrier the
    // extra waves cost more than the issue stalls they remove.  Kept as a tested configuration (bit-identical) and behind UNI_H2D_SPEC=1.
    static const bool want_spec = getenv("UNI_H2D_SPEC") != nullptr;
    const bool plain = !a.stats && a.splitk <= 1;
    const bool spec = plain && (cfg == 422 || cfg == 423 || (want_spec && a.force_cfg % 1000 == 0 && (cfg == 322 || cfg == 323)));
    if (cfg == 422) cfg = 322;
    if (cfg == 423) cfg = 323;
#define GOS(WM, WN, TM, TN, NST) return conv ? launch_h2d_spec<WM, WN, TM, TN, true, NST>(a, s) : launch_h2d_spec<WM, WN, TM, TN, false, NST>(a, s)
    if (spec && cfg == 322) { GOS(2, 2, 2, 2, 4); }
    if (spec && cfg == 323) { GOS(4, 1, 1, 3, 4); }
#undef GOS
    switch (cfg) {
        case 322: GOD(2, 2, 2, 2, 4);
        case 323: GOD(4, 1, 1, 3, 4);
        case 332: GOD(2, 2, 1, 2, 3);
        case 346: GOD(4, 2, 2, 3, 2);     // 256 x 192, 8 waves, 2 slots
        default: GOD(2, 2, 1, 1, 3);
    }
#undef GOD
}
