// PING-PONG persistent 256 x 256 split-f16 ("f16x2") GEMM for the plain contractions: ConvNeXt
// pointwise MLPs and transformer Linears (convnext.py:41-54 pwconv1 / pwconv2; deformable_transformer.py:122-131).
//
// 8 waves = 2 wave groups (M halves) x 4 (N quarters), a wave owns 128 x 64 of the tile (128 accumulator registers).  A K step
// (32 k = 128-byte LDS rows of [8 hi][8 lo] groups) is cut into four HALF-TILES of 128 rows -- A0, A1 (the first / second 64 rows
// of both groups' M halves), B0, B1 (the first / second 32 columns of every wave's N quarter) -- and into four PHASES, one
// 64 x 32 quadrant of the wave tile each:
//      phase 1: reads A0 + B0, quadrant (A0, B0)        phase 3: reads A1, quadrant (A1, B1)
//      phase 2: reads B1,      quadrant (A0, B1)        phase 4: (B0 kept), quadrant (A1, B0)
// A phase is [ds_reads and / or two half-tiles of LDS-DMA] | barrier | [12 MFMAs] | barrier, and the second wave group runs ONE BARRIER
// BEHIND the first: while one group issues its MFMAs, the other one (same SIMDs) reads fragments and requests DMA, so the
// matrix pipe never waits for LDS.  The DMA stream runs continuously over the K steps AND over the tiles of the persistent block,
// ~2 steps ahead in two 64-KiB stages, drained by ONE counted `s_waitcnt vmcnt(6)` per K step (three half-tiles stay in flight
// across every barrier, raw `s_barrier` only):
//      half-tile:   B0(S+1)   A0(S+2)   B1(S+2)   A1(S+2)       requested in phase 2 / 2 / 4 / 4 of step S (phases 1 and 3 carry the 12 + 8
//      last read:   p1(S-1)   p1(S)     p2(S)     p3(S)         fragment reads of A0 + B0 / A1 and no request) -- each at least ONE phase after the
// last read of the slot it overwrites (reads are retired with
// lgkmcnt(0) before the barrier that ends their phase); after the wait in phase 4 every half-tile of step S+1 has landed for all
// waves once both groups have passed their next barrier, i.e. before anybody's phase 1 of step S+1.
// Epilogue (bias, activation, residual, fp32 / operand-format outputs through a per-wave 4-KiB
// staging block), staged in its own 32 KiB so the DMA of the next tile keeps flying.
#include "kernels.h"
#include "gemm_epi.h"

// GemmArgs::dbg (tools/gemm_h2_bench.py, cfg + 1000 * bits): 1 = no DMA requests, 2 = no MFMAs, 4 = no epilogue stores, 16 = no epilogue

namespace {
constexpr int NW = 8;
constexpr int BM = 256, BN = 256, BKE = 32, ROWB = 128;
constexpr int HALF = 128 * ROWB;                 // one half-tile: 128 rows x 128 B
constexpr int STAGE = 4 * HALF;                  // A0 A1 B0 B1 of one K step
constexpr int OPER = 2 * STAGE;
constexpr int STG = 4096;                        // per-wave epilogue staging block
constexpr int LDS_BYTES = OPER + NW * STG;       // 163840 = all of the CU's LDS

struct Tile { int m0, n0, k0, split; };      // k0: first K step of the unit (split-K), split: index of its K range
// A side of a tile: buffer descriptor + per-lane offsets [half][piece]; CONV: byte offset (< 2^26, launcher checks) | validity bits << 26.
// (At namespace scope: a struct local to the kernel template makes hipcc drop the host stubs of its instantiations.)
struct ATile { __amdgpu_buffer_rsrc_t rs; int vo[2][2]; };
// unit u of a split-K launch = (K range u / ntiles, tile u % ntiles): neighbouring units are different tiles of the same K range (shared panels)
__device__ __forceinline__ Tile tile_of(int u, int nbm, int nbn, int nku) {
    const int ntiles = nbm * nbn;
    const int split = u / ntiles, L = u - split * ntiles;
    constexpr int GN = 8;           // N is cut into chunks of 8 tiles; inside a chunk tiles run M-major (see gemm.hip)
    const int per_chunk = nbm * GN;
    const int c = L / per_chunk;
    const int wc = min(GN, nbn - c * GN);
    const int rem = L - c * per_chunk;
    const int bm = rem / wc;
    return {bm * BM, (c * GN + rem - bm * wc) * BN, split * nku, split};
}
__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// raw workgroup barrier that neither the IR optimiser nor the machine scheduler moves anything across
__device__ __forceinline__ void phase_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
}
}  // namespace

#define Q_WAIT_LDS() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define Q_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// FMT = FMT_H2: split-f16 operands, 32 k per step, 3 MFMAs per product; FMT = FMT_BF16: the bf16 twin on the same schedule -- 64 k per
// step in the same 128-byte rows, 4 k slices, 1 MFMA per product (its 8-MFMA phases are shorter than the LDS/DMA phases: that
// mode is bound by the L2 -> LDS path, not by the matrix pipe)
template <int ACT, bool OUTF, bool CONV, bool STATS, int FMT>
__global__ __launch_bounds__(64 * NW) void gemm_h2q_kernel(GemmArgs p) {
    constexpr int EB = FMT == FMT_H2 ? 4 : 2;          // bytes per operand element
    constexpr int KS = ROWB / EB;                      // k per step: 32 (f16x2) or 64 (bf16)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wc = wave & 3;
    const int fr = lane & 31, fh = lane >> 5;

    const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
    int first, stride, count;
    {
        const int ntiles = nbm * nbn * (p.splitk > 1 ? p.splitk : 1);      // work units
        const int x = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
        const int q = ntiles >> 3, r = ntiles & 7;
        const int start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
        const int cnt = q + (x < r ? 1 : 0);
        first = start + slot;
        stride = nslots;
        count = slot < cnt ? (cnt - slot + nslots - 1) / nslots : 0;
    }
    if (count == 0) return;

    // ---- DMA: wave w fills pieces w and w + 8 (8 rows = 1 KiB each) of every half-tile; lane -> row lane / 8 of the piece, 16-byte
    // chunk (lane & 7) of the LDS row, which holds the row's LOGICAL chunk lch (source-side swizzle with (LDS row >> 1) & 7).
    // `buffer_load_dwordx4 ... lds` through one descriptor per operand and tile: base = first row of the tile, num_records = its
    // valid rows (rows past M read as zeros, no clamp), per-lane offsets are the same for every tile and K step, the K step
    // travels in the scalar offset.
    const int lrow = lane >> 3;
    const int lch = (lane & 7) ^ ((4 * (wave & 1) + (lrow >> 1)) & 7);
    const int lda4 = p.lda * EB, ldw4 = p.Kpad * EB;   // row strides in bytes
    int vob[2][2];                                   // [half][piece]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) vob[h][i] = ((wave >> 2) * 64 + 8 * (wave & 3) + lrow + 128 * i + 32 * h) * ldw4 + lch * 16;     // tile column
    const int arow = 8 * wave + lrow;                // tile row of piece i, half h: arow + 128 i + 64 h
    // A side of one tile.  Plain GEMM: the descriptor starts at the tile's first row, per-lane offsets are tile independent.
    // Implicit GEMM (CONV; Cin % 32 == 0, so a K step = 32 channels of ONE tap): the descriptor starts (pad, pad) pixels before the
    // tile's first input pixel, the per-lane offset is the lane's pixel relative to that, the tap (ky, kx) and the channel block
    // travel in the scalar offset ((ky Win + kx) lda + c) -- never negative -- and a lane whose tap falls outside the image (or
    // whose row is past M) presents an offset beyond num_records: the buffer load returns zeros, which IS the zero padding.
    auto make_a = [&](Tile tl, ATile& c) __attribute__((always_inline)) {
        char* abase = const_cast<char*>(reinterpret_cast<const char*>(p.A));
        if (!CONV) {
            const int rows = min(BM, p.M - tl.m0);
            c.rs = __builtin_amdgcn_make_buffer_rsrc(abase + (long)tl.m0 * lda4, 0, rows * lda4, 0x00020000);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 2; ++i) c.vo[h][i] = (arow + 128 * i + 64 * h) * lda4 + lch * 16;
        } else {
            const int b0 = tl.m0 / p.Mper, q0 = tl.m0 - b0 * p.Mper, oy0 = q0 / p.Wout, ox0 = q0 - oy0 * p.Wout;
            const int pix0 = (b0 * p.Hin + oy0 * p.stride) * p.Win + ox0 * p.stride;
            c.rs = __builtin_amdgcn_make_buffer_rsrc(abase + ((long)pix0 - (p.pad * p.Win + p.pad)) * lda4, 0, 0x7ffffff0, 0x00020000);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int m = tl.m0 + arow + 128 * i + 64 * h;
                    const bool ok = m < p.M;
                    const int mm = ok ? m : tl.m0;
                    const int b = mm / p.Mper, q = mm - b * p.Mper, oy = q / p.Wout, ox = q - oy * p.Wout;
                    const int off = ((b * p.Hin + oy * p.stride) * p.Win + ox * p.stride - pix0) * lda4 + lch * 16;
                    int bits = 0;                    // bit ky: input row of tap row ky inside the image; bit 3 + kx: the same for the column
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        const int iy = oy * p.stride - p.pad + t, ix = ox * p.stride - p.pad + t;
                        bits |= (t < p.KH && iy >= 0 && iy < p.Hin) ? (1 << t) : 0;
                        bits |= (t < p.KW && ix >= 0 && ix < p.Win) ? (8 << t) : 0;
                    }
                    c.vo[h][i] = off | ((ok ? bits : 0) << 26);
                }
        }
    };
    auto rsrc_b = [&](Tile tl) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.W)) + (long)tl.n0 * ldw4, 0, BN * ldw4, 0x00020000);
    };
    const int cs = CONV ? p.Cin / KS : 1;            // K steps per tap
    const int csm = 65536 / cs + 1;                  // fk / cs = (fk * csm) >> 16 for fk < 9 cs <= 2^12 (launcher checks)
    const int kwm = p.KW == 3 ? 11 : p.KW == 2 ? 16 : 32;   // tap / KW = (tap * kwm) >> 5 for tap < 9
#define Q_DMA(rs, vo, soff, dst) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst), 16, vo, soff, 0, 0)
    auto issueA = [&](int h, int par, int kt, const ATile& c) __attribute__((always_inline)) {
        char* dst = smem + par * STAGE + h * HALF + wave * 1024;
        if (!CONV) {
            Q_DMA(c.rs, c.vo[h][0], kt * ROWB, dst);
            Q_DMA(c.rs, c.vo[h][1], kt * ROWB, dst + 8192);
        } else {
            const int tap = (kt * csm) >> 16, cstep = kt - tap * cs;
            const int ky = (tap * kwm) >> 5, kx = tap - ky * p.KW;
            const int soff = (ky * p.Win + kx) * lda4 + cstep * ROWB;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bool ok = ((c.vo[h][i] >> (26 + ky)) & (c.vo[h][i] >> (29 + kx)) & 1) != 0;
                Q_DMA(c.rs, ok ? (c.vo[h][i] & 0x03ffffff) : 0x7fffffff, soff, dst + i * 8192);
            }
        }
    };
    auto issueB = [&](int h, int par, int kt, __amdgpu_buffer_rsrc_t rs) __attribute__((always_inline)) {
        char* dst = smem + par * STAGE + (2 + h) * HALF + wave * 1024;
        Q_DMA(rs, vob[h][0], kt * ROWB, dst);
        Q_DMA(rs, vob[h][1], kt * ROWB, dst + 8192);
    };

    // ---- fragment reads: group `grp` reads rows 64 grp + 32 i + fr of an A half, wave column wc rows 32 wc + fr of a B half;
    // per lane four chunk addresses (hi / lo of k slice 0 / 1), everything else is an immediate offset
    const int sw = (fr >> 1) & 7;
    // f16x2: chunks (hi, lo) of k slice 0, (hi, lo) of k slice 1 = 2 fh, 2 fh + 1, 4 + 2 fh, 5 + 2 fh;  bf16: k slices 0..3 = fh, 2 + fh, 4 + fh, 6 + fh
    const int q0 = FMT == FMT_H2 ? 2 * fh : fh, q1 = FMT == FMT_H2 ? 2 * fh + 1 : 2 + fh;
    const int q2 = FMT == FMT_H2 ? 4 + 2 * fh : 4 + fh, q3 = FMT == FMT_H2 ? 5 + 2 * fh : 6 + fh;
    const int a_rd = (grp * 64 + fr) * ROWB, b_rd = 2 * HALF + (wc * 32 + fr) * ROWB;
    const int lds0 = (int)(size_t)(__attribute__((address_space(3))) char*)smem;
    if (lds0 != 0) __builtin_trap();                 // the stage toggle below XORs bit 16 of the addresses: the dynamic LDS block must start at 0
    int ra4[4] = {lds0 + a_rd + ((q0 ^ sw) << 4), lds0 + a_rd + ((q1 ^ sw) << 4), lds0 + a_rd + ((q2 ^ sw) << 4), lds0 + a_rd + ((q3 ^ sw) << 4)};
    int rb4[4] = {lds0 + b_rd + ((q0 ^ sw) << 4), lds0 + b_rd + ((q1 ^ sw) << 4), lds0 + b_rd + ((q2 ^ sw) << 4), lds0 + b_rd + ((q3 ^ sw) << 4)};
#define Q_LDS(addr, off) (*reinterpret_cast<const __attribute__((address_space(3))) f16x8*>((size_t)((addr) + (off))))

    f32x16 acc[4][2];
    // fragment registers [.][0..3]: f16x2 = (hi, lo) of k slice 0, (hi, lo) of k slice 1; bf16 = k slices 0..3.  B fragments of both
    // halves stay in registers (B0 serves phases 1 and 4)
    f16x8 fa[2][4], fb[2][4];
    const int nk = (p.K / KS) / (p.splitk > 1 ? p.splitk : 1);       // K steps per unit (the launcher makes the split divide the K steps)
    Tile cur = tile_of(first, nbm, nbn, nk);
    Tile nxt = count > 1 ? tile_of(first + stride, nbm, nbn, nk) : cur;
    ATile ta_cur, ta_nxt;
    make_a(cur, ta_cur);
    make_a(nxt, ta_nxt);
    __amdgpu_buffer_rsrc_t rb_cur = rsrc_b(cur), rb_nxt = rsrc_b(nxt);

#define Q_LD_A(h)                                                                                   \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int q = 0; q < 4; ++q)      \
        fa[i][q] = Q_LDS(ra4[q], (h) * HALF + i * 32 * ROWB);
#define Q_LD_B(h) _Pragma("unroll") for (int q = 0; q < 4; ++q) fb[h][q] = Q_LDS(rb4[q], (h) * HALF);
    // quadrant (HA, HB): accumulator tiles (2 HA + i, HB); per k slice the two cross terms first, then hi.hi
#define Q_MF16(b, a, c) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c, 0, 0, 0)
#define Q_MB16(b, a, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a), c, 0, 0, 0)
#define Q_MFMA(HA, HB)                                                                                   \
    if (FMT == FMT_H2) {                                                                                 \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                               \
            Q_MF16(fb[HB][2 * kk + 1], fa[0][2 * kk], acc[2 * (HA)][HB]);                                \
            Q_MF16(fb[HB][2 * kk + 1], fa[1][2 * kk], acc[2 * (HA) + 1][HB]);                            \
            Q_MF16(fb[HB][2 * kk], fa[0][2 * kk + 1], acc[2 * (HA)][HB]);                                \
            Q_MF16(fb[HB][2 * kk], fa[1][2 * kk + 1], acc[2 * (HA) + 1][HB]);                            \
            Q_MF16(fb[HB][2 * kk], fa[0][2 * kk], acc[2 * (HA)][HB]);                                    \
            Q_MF16(fb[HB][2 * kk], fa[1][2 * kk], acc[2 * (HA) + 1][HB]);                                \
        }                                                                                                \
    } else {                                                                                             \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                               \
            Q_MB16(fb[HB][kk], fa[0][kk], acc[2 * (HA)][HB]);                                            \
            Q_MB16(fb[HB][kk], fa[1][kk], acc[2 * (HA) + 1][HB]);                                        \
        }                                                                                                \
    }
    // (k step, tile) `ahead` steps after the current one: the stream continues into the block's next tile
#define Q_FUT(ahead)                                                   \
    int fk = kt + (ahead);                                             \
    const bool fnx = fk >= nk;                                         \
    fk -= fnx ? nk : 0;                                                \
    fk += fnx ? nxt.k0 : cur.k0;              /* absolute K step: the unit's K range starts at k0 */ \
    const __amdgpu_buffer_rsrc_t fb = fnx ? rb_nxt : rb_cur;                                  \
    ATile fa;                                                                                  \
    fa.rs = fnx ? ta_nxt.rs : ta_cur.rs;                                                       \
    _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {  \
        fa.vo[h_][i_] = (CONV && fnx) ? ta_nxt.vo[h_][i_] : ta_cur.vo[h_][i_];                 \
    }

    // ---- prologue: step 0 completely, step 1 without its B0 (phase 1 of step 0 requests that one)
    {
        const int kt = 0;
        issueA(0, 0, cur.k0, ta_cur); issueB(0, 0, cur.k0, rb_cur); issueB(1, 0, cur.k0, rb_cur); issueA(1, 0, cur.k0, ta_cur);
        Q_FUT(1);
        issueA(0, 1, fk, fa); issueB(1, 1, fk, fb); issueA(1, 1, fk, fa);
    }
    Q_WAIT_VM(6);
    phase_barrier();
    if (grp == 1) phase_barrier();                    // the second group runs one barrier behind
    int par = 0;
#pragma unroll 1
    for (int t = 0; t < count; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll 1
        for (int kt = 0; kt < nk; ++kt) {
            {   // phase 1
                Q_LD_A(0); Q_LD_B(0);                 // 12 reads: no DMA request in this phase
                Q_WAIT_LDS();
                phase_barrier();
                if (!(p.dbg & 2)) { Q_MFMA(0, 0); }
                phase_barrier();
            }
            {   // phase 2
                Q_LD_B(1);
                if (!(p.dbg & 1)) { Q_FUT(1); issueB(0, par ^ 1, fk, fb); }     // B0(S+1) first: it is the oldest request the phase-4 wait must retire
                if (!(p.dbg & 1)) { Q_FUT(2); issueA(0, par, fk, fa); }
                Q_WAIT_LDS();
                phase_barrier();
                if (!(p.dbg & 2)) { Q_MFMA(0, 1); }
                phase_barrier();
            }
            {   // phase 3
                Q_LD_A(1);
                Q_WAIT_LDS();
                phase_barrier();
                if (!(p.dbg & 2)) { Q_MFMA(1, 1); }
                phase_barrier();
            }
            {   // phase 4 (B0 is still in registers)
                if (!(p.dbg & 1)) { Q_FUT(2); issueB(1, par, fk, fb); issueA(1, par, fk, fa); }
                Q_WAIT_VM(6);
                Q_WAIT_LDS();
                phase_barrier();
                if (!(p.dbg & 2)) { Q_MFMA(1, 0); }
                phase_barrier();
            }
            par ^= 1;
#pragma unroll
            for (int q = 0; q < 4; ++q) { ra4[q] ^= STAGE; rb4[q] ^= STAGE; }
        }
        // ---- tile done (the DMA stream is already inside the next tile)
        const Tile done = cur;
        cur = nxt; ta_cur = ta_nxt; rb_cur = rb_nxt;
        if (t + 2 < count) { nxt = tile_of(first + (t + 2) * stride, nbm, nbn, nk); make_a(nxt, ta_nxt); rb_nxt = rsrc_b(nxt); }
        char* st = smem + OPER + wave * STG;
        const int nw0 = done.n0 + wc * 64;
        const int rb0 = done.m0 + grp * 128;
        const int srow = p.splitk > 1 ? done.split * p.M : 0;     // split-K: outF is the slab [splitk][M][N], this unit's K range owns slab `split`
        if (p.dbg & 16) {                             // ablation: no drain, accumulators kept live
            float sacc = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
            if (sacc == 1.2345e-30f && p.outF) p.outF[0] = sacc;
            continue;
        }
        if (STATS) {
            // GroupNorm-statistics problems end with a block-wide reduction: the first wave group waits one barrier for the
            // second one (which runs one barrier behind), both drain in step, and the second group re-staggers afterwards
            if (grp == 0) phase_barrier();
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] *= p.wscale;
        }
        const float ws = STATS ? 1.f : p.wscale;
        // bias of the wave's 64 columns through SCALAR loads (constant address space): SMEM does not sit on the vector-memory
        // counter, so no wait of the epilogue drains the DMA stream or the stores behind it.  Lane (fr, fh) owns columns 8 g + 4 fh + e.
        const bool has_res = OUTF && !STATS && !CONV && p.res != nullptr, has_b = p.outB != nullptr;
#pragma unroll
        for (int j = 0; j < 2; ++j) {                  // 32-column slab of the wave tile: its bias once, then the four 32-row blocks
            float bv[16];
            {
                typedef float f32x8 __attribute__((ext_vector_type(8)));
                const __attribute__((address_space(4))) float* bias = (const __attribute__((address_space(4))) float*)(p.bias);
                if (!p.bias) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) bv[r] = 0.f;
                } else if (nw0 + 64 <= p.N) {                   // uniform
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x8 b8 = *reinterpret_cast<const __attribute__((address_space(4))) f32x8*>(bias + nw0 + j * 32 + 8 * g);
#pragma unroll
                        for (int e = 0; e < 4; ++e) bv[4 * g + e] = fh ? b8[4 + e] : b8[e];
                    }
                } else {                                        // ragged last N tile: clamped element loads
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int c = nw0 + j * 32 + 8 * g + e;
                            const float lo = bias[min(c, p.N - 1)], hi = bias[min(c + 4, p.N - 1)];
                            bv[4 * g + e] = fh ? hi : lo;
                        }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // residual rows of this 32 x 32 block requested first: their latency hides under the activation + staging below
                f32x4 rv[4];
                if (has_res) {
                    const int c = lane & 7, rr = lane >> 3;
                    const int col = nw0 + j * 32 + 4 * c;
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        const int row = rb0 + i * 32 + tt * 8 + rr;
                        const bool ok = row < p.M && col < p.N;
                        rv[tt] = *reinterpret_cast<const f32x4*>(p.res + (ok ? (size_t)row * p.ldr + col : 0));
                    }
                }
                // fp32 [32 rows][32 cols] of block (i, j): 128-B rows, float4 chunks XOR-swizzled with row & 7
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_fast<ACT>(fmaf(acc[i][j][4 * g + e], ws, bv[4 * g + e]));
                    *reinterpret_cast<f32x4*>(st + fr * 128 + (((2 * g + fh) ^ (fr & 7)) << 4)) = v;
                }
                wave_fence();
                if (OUTF) {        // fp32 (+ residual) output, optional operand-format copy: 4 channels per lane, 8 rows per instruction
                    const int c = lane & 7, rr = lane >> 3;
                    const int col = nw0 + j * 32 + 4 * c;
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        const int r = tt * 8 + rr, row = rb0 + i * 32 + r;
                        f32x4 v = *reinterpret_cast<const f32x4*>(st + r * 128 + ((c ^ (r & 7)) << 4));
                        if (has_res) v += rv[tt];
                        if (row < p.M && col < p.N && !(p.dbg & 4)) {
                            const int orow = (p.out_hw ? (row / p.out_hw) * p.out_stride + p.out_off + row % p.out_hw : row) + srow;
                            *reinterpret_cast<f32x4*>(p.outF + (size_t)orow * p.ldf + col) = v;
                            if (has_b) act_store4(p.outB, (size_t)row * p.ldb + col, v[0], v[1], v[2], v[3], FMT);
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
                        if (row < p.M && col < p.N && !(p.dbg & 4)) {
                            const float v8[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                            // streaming (non-temporal) stores: the consumer is the next launch and the tensor (786 MB for a stage-2
                            // pwconv1) is far larger than L2, keeping it there only evicts the operand panels of this GEMM
                            if (FMT == FMT_H2) {
                                f16x8 h, l;
#pragma unroll
                                for (int q = 0; q < 8; ++q) { f16 hh, ll; h2_split(v8[q], hh, ll); h[q] = hh; l[q] = ll; }
                                char* op = reinterpret_cast<char*>(p.outB) + ((size_t)row * p.ldb + col) * 4;
                                __builtin_nontemporal_store(__builtin_bit_cast(f32x4, h), reinterpret_cast<f32x4*>(op));
                                __builtin_nontemporal_store(__builtin_bit_cast(f32x4, l), reinterpret_cast<f32x4*>(op + 16));
                            } else {
                                bf16x8 o;
#pragma unroll
                                for (int q = 0; q < 8; ++q) o[q] = (bf16)v8[q];
                                __builtin_nontemporal_store(__builtin_bit_cast(f32x4, o), reinterpret_cast<f32x4*>(p.outB + (size_t)row * p.ldb + col));
                            }
                        }
                    }
                }
                wave_fence();
            }
        }
        if (STATS) {
            gemm_stats<2, 4, 4, 2>(p, acc, done.m0, done.n0, grp, wc, lane, tid, reinterpret_cast<float*>(smem + OPER));
            __syncthreads();                          // the reduction scratch is the waves' staging blocks
            if (grp == 1) phase_barrier();
        }
    }
    Q_WAIT_VM(0);                                     // the stream's last requests (never consumed) must not outlive the block's LDS
    if (grp == 0) phase_barrier();                    // barrier count of the two groups evens out
}

template <int ACT, bool OUTF, bool CONV, bool STATS, int FMT>
static int launch_h2q_inst(const GemmArgs& a, int grid, hipStream_t s) {
    static DevOnce attr_once;
    UNI_LDS_OPTIN(attr_once, "gemm_h2q", LDS_BYTES, reinterpret_cast<const void*>(&gemm_h2q_kernel<ACT, OUTF, CONV, STATS, FMT>));
    hipLaunchKernelGGL((gemm_h2q_kernel<ACT, OUTF, CONV, STATS, FMT>), dim3(grid), dim3(64 * NW), LDS_BYTES, s, a);
    return 0;
}

// plain GEMMs: what the older persistent kernels take minus the LayerNorm fold; implicit GEMMs (3x3 / strided convs) and
// GroupNorm-statistics problems: step-aligned taps (Cin a multiple of the 32 / 64 k of a step), fp32 output, no activation window
bool gemm_h2q_supported(const GemmArgs& a) {
    const bool conv = a.KH != 1 || a.KW != 1 || a.stride != 1 || a.pad != 0;
    if (a.b32 != FMT_H2 && a.b32 != FMT_BF16) return false;
    const int ks = a.b32 == FMT_H2 ? 32 : 64, eb = a.b32 == FMT_H2 ? 4 : 2;
    if (!a.epi || a.K % ks != 0 || a.K < 2 * ks || a.act_col0 != 0) return false;
    if (!conv && !a.stats) {
        if (a.b32 != FMT_H2) return gemm_p44_supported(a);
        return a.out_hw == 0 && (a.act == ACT_NONE || a.act == ACT_RELU || a.act == ACT_GELU) && (a.outF || a.outB) && (a.outF || !a.res);   // (bias optional)
    }
    if (conv) {
        if (a.Cin % ks != 0 || a.KH > 3 || a.KW > 3 || a.K != a.KH * a.KW * a.Cin || a.Cin / ks > 448) return false;
        // the input pixels of one 256-row tile must lie within 2^26 bytes of its first one (the per-lane offsets carry 6 flag bits)
        const long span_pix = ((long)BM / a.Wout + 2) * a.stride * a.Win + (long)BM * a.stride + 3L * a.Win;
        if (span_pix * a.lda * eb >= (1L << 26)) return false;
    }
    if (a.res) return false;                         // residual adds only on the plain path
    if (a.stats) return a.act == ACT_NONE && a.outF != nullptr && a.cpg > 0 && 256 / a.cpg + 2 <= 64;
    return a.outF != nullptr ? (a.act == ACT_NONE || a.act == ACT_RELU) : (a.outB != nullptr && !a.res && !a.out_hw && (a.act == ACT_NONE || a.act == ACT_RELU));
}

template <int FMT>
static int launch_h2q_fmt(const GemmArgs& a, int grid, hipStream_t s) {
    const bool f = a.outF != nullptr;
    const bool conv = a.KH != 1 || a.KW != 1 || a.stride != 1 || a.pad != 0;
    if (a.stats) return conv ? launch_h2q_inst<ACT_NONE, true, true, true, FMT>(a, grid, s) : launch_h2q_inst<ACT_NONE, true, false, true, FMT>(a, grid, s);
    if (conv) {
        if (a.act == ACT_RELU) return f ? launch_h2q_inst<ACT_RELU, true, true, false, FMT>(a, grid, s) : launch_h2q_inst<ACT_RELU, false, true, false, FMT>(a, grid, s);
        return f ? launch_h2q_inst<ACT_NONE, true, true, false, FMT>(a, grid, s) : launch_h2q_inst<ACT_NONE, false, true, false, FMT>(a, grid, s);
    }
    switch (a.act) {
        case ACT_GELU: return f ? launch_h2q_inst<ACT_GELU, true, false, false, FMT>(a, grid, s) : launch_h2q_inst<ACT_GELU, false, false, false, FMT>(a, grid, s);
        case ACT_RELU: return f ? launch_h2q_inst<ACT_RELU, true, false, false, FMT>(a, grid, s) : launch_h2q_inst<ACT_RELU, false, false, false, FMT>(a, grid, s);
        default: return f ? launch_h2q_inst<ACT_NONE, true, false, false, FMT>(a, grid, s) : launch_h2q_inst<ACT_NONE, false, false, false, FMT>(a, grid, s);
    }
}

int launch_gemm_h2q(const GemmArgs& a, hipStream_t s) {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
        ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        ncu -= ncu % 8;
    }
    const int sk = a.splitk > 1 ? a.splitk : 1;
    const int ntiles = cdiv(a.M, BM) * cdiv(a.N, BN) * sk;
    const int grid = ntiles < ncu ? (ntiles + 7) / 8 * 8 : ncu;
    if (sk > 1) {     // K ranges -> slab [sk][M][N]; bias / residual / GroupNorm sums are applied by the reduce kernel (gemm_h2.hip)
        GemmArgs g = gemm_splitk_partial_args(a);
        int rc = launch_h2q_fmt<FMT_H2>(g, grid, s);
        return rc ? rc : launch_splitk_reduce(a, s);
    }
    return a.b32 == FMT_H2 ? launch_h2q_fmt<FMT_H2>(a, grid, s) : launch_h2q_fmt<FMT_BF16>(a, grid, s);
}
