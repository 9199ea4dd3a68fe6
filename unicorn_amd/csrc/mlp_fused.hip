// Fused ConvNeXt MLP for the narrow stages (convnext.py:47-54, Block.forward after the LayerNorm):
//
//     out[m][:] = res[m][:] + b2 + W2' . GELU(W1 . a[m][:] + b1)            W2' = diag(gamma) W2, 4C hidden units
//
// in ONE launch, "f16x2" operands (hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16, fp32 accumulate; common.h FMT_H2).  The
// unfused pair (gemm_h2q: pwconv1 + GELU -> hidden tensor -> pwconv2 + residual) spends more time on the 4-byte-per-element hidden
// tensor than in its K loops when C is small: stage 0 of the large model (M = 1 024 000 rows per 16 frames, C = 192) writes and
// re-reads 3.1 GB per block and runs at 210 / 262 TFLOP/s-equivalent.  Here the hidden activations never leave the registers:
//
//  * 4 waves per block, one per SIMD (512-register budget), each wave OWNS 32 pixel rows of a 128-row tile for both GEMMs.
//    Lane (fr, fh) = (lane & 31, lane >> 5) holds row fr.  The A operand (32 x C, f16x2) is loaded once per tile straight into
//    MFMA fragments (C / 2 registers) -- nothing to share between waves, so it never touches LDS.
//  * GEMM1 (swapped operands: weights as the MFMA "A", pixels as "B") leaves lane (fr, fh) with hidden units 8g + 4fh + e
//    (g, e < 4) of a 32-unit block = exactly what a B-operand fragment of GEMM2 needs IF the k order inside every group of 16
//    hidden units is [0-3, 8-11 | 4-7, 12-15]: the host packs W2 with that permutation, so bias + GELU + hi/lo split of the 16
//    accumulators ARE the GEMM2 operand (the flash-attention P -> PV chaining, with GELU in the place of the softmax).
//  * Only the weights go through LDS: the host lays W1 / W2 out as a stream of 2 * NH "pieces" (NH = 4C / 32 hidden blocks; a
//    piece = the 32 x C slab of W1 or the C x 32 slab of W2 of one hidden block, 128 C bytes) in consumption order, each piece
//    already in its bank-conflict-free LDS image, so the fill is a linear `buffer_load_dwordx4 ... lds` stream into a ring of
//    3-6 slots, NSLOT - 1 pieces ahead of the MFMAs, one counted `s_waitcnt vmcnt` + one raw `s_barrier` per piece.
//  * Software pipeline inside a tile: GEMM1 of hidden block h + 1 is issued beside the GELU of block h (VALU under MFMA), then
//    GEMM2 of block h.  Two accumulator chains in GEMM1 (even / odd k slices), C / 32 chains in GEMM2.
//
// Work per 128-row tile and wave: 2 * NH * 3 C / 16 MFMAs; LDS reads 2 x 16 B per 3 MFMAs; DMA 32 C^2 bytes per tile and block
// (C = 192: 1.18 MB per 55k MFMA cycles = 11 B/clk/CU, half of what the 256 x 256 GEMM tiles stream).
#include "kernels.h"

#include <cstring>
#include <type_traits>
#include <vector>

namespace {
constexpr int MW = 4;                    // waves per block
constexpr int BMF = 32 * MW;             // rows per tile

template <int C>
struct Geo {
    static constexpr int NS = C / 16;                // k slices of the A operand
    static constexpr int NH = C / 8;                 // hidden blocks of 32 units (4C / 32)
    static constexpr int NJ = C / 32;                // output column blocks
    static constexpr int PB = 128 * C;               // bytes per weight piece
    static constexpr int IPW = C / 32;               // 1-KiB DMA instructions per wave and piece
    static constexpr int BIAS = 20 * C;              // b1 (4C floats) + b2 (C floats) resident in LDS behind the ring
    static constexpr int NSLOT = ((163840 - BIAS) / PB) < 6 ? ((163840 - BIAS) / PB) : 6;
    static constexpr int LDS = NSLOT * PB + BIAS;
    static constexpr int NP = 2 * NH;                // pieces per tile
    static constexpr int KW = (NSLOT - 2) * IPW;     // DMA instructions of this wave that may stay in flight behind the piece being waited for
    static_assert(C % 32 == 0 && NSLOT >= 3 && KW <= 60, "geometry");
};

__device__ __forceinline__ void raw_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
#define F_LDS(ptr) (*reinterpret_cast<const __attribute__((address_space(3))) f16x8*>(ptr))
#define F_MFMA(w, a, c) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, a, c, 0, 0, 0)
}  // namespace

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// CH = accumulator chains of GEMM1 (2: even / odd k slices; 1 where the registers do not allow the second one)
template <int C, int CH, bool OUTB, int DBG>
__global__ __launch_bounds__(64 * MW, 1) void mlp_fused_kernel(MlpArgs p) {
    using G = Geo<C>;
    constexpr int NS = G::NS, NH = G::NH, NJ = G::NJ, PB = G::PB, IPW = G::IPW, NSLOT = G::NSLOT, NP = G::NP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;

    const int ntiles = (p.M + BMF - 1) / BMF;
    const int count = ((int)blockIdx.x < ntiles) ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    if (count == 0) return;
    const int total = count * NP;                    // pieces this block consumes

    // ---- weight stream: piece P of the block comes from blob + (P % NP) * PB, goes to ring slot P % NSLOT.  Requests past the
    // block's last piece keep the stream's shape (same vmcnt arithmetic, no branch): their offset lies beyond num_records, so they
    // write zeros into a slot nobody reads any more.
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.blob), 0, NP * PB, 0x00020000);
    const int vo_w = lane * 16 + wave * 1024;
    const int lds0 = (int)(size_t)(__attribute__((address_space(3))) char*)smem;
    // Blob order (mlp_pack_host): q = 2 i -> W1_i, q = 2 i + 1 -> W2_{i-1 mod NH}.  A block starts its tiles at hidden block h0 and consumes
    // W1_{h0} | W1_{h0+1} W2_{h0} | ... | W1_{h0-1} W2_{h0-2} | W2_{h0-1}  (indices mod NH) = blob positions 2 h0, then 2 h0 + 2, 2 h0 + 3, ...
    // cyclically, and 2 h0 + 1 last.  The blocks of one XCD (blockIdx & 7 fixed) spread their h0 over the stream: all CUs reading
    // the same 24 KB at the same time would sit on the few L2 channels that hold it (4-KiB channel interleave).
    const int h0 = (p.dbg & 8) ? 0 : (int)(((blockIdx.x >> 3) * 13u + (blockIdx.x & 7) * 5u) % NH);
    const int q_first = 2 * h0, q_last = 2 * h0 + 1;
    auto hb_of = [&](int step) __attribute__((always_inline)) { const int v = h0 + step; return v >= NH ? v - NH : v; };
    int p_issue = 0, src_q = q_first, wr_slot = 0;   // next piece to request: its index, its position in the blob, its slot
    auto issue_piece_part = [&](auto II) __attribute__((always_inline)) {      // request II of the IPW of the next piece
        constexpr int i = decltype(II)::value;
        const int so = (p_issue < total && !(DBG & 1)) ? src_q * PB : 0x40000000;
        char* dst = smem + wr_slot * PB + wave * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(dst + i * 4096), 16, vo_w, so + i * 4096, 0, 0);
    };
    auto issue_piece_done = [&]() __attribute__((always_inline)) {
        ++p_issue;
        if (src_q == q_last) src_q = q_first;                        // next tile
        else {
            int n = src_q == q_first ? src_q + 2 : src_q + 1;
            n = n >= NP ? n - NP : n;
            src_q = n == q_first ? q_last : n;                       // the cycle is complete: the piece held back comes last
        }
        wr_slot = wr_slot + 1 == NSLOT ? 0 : wr_slot + 1;
    };
    auto issue_piece = [&]() __attribute__((always_inline)) {
        static_for<0, IPW>([&](auto II) __attribute__((always_inline)) { issue_piece_part(II); });
        issue_piece_done();
    };
    int rd_slot = 0;
    // start of the piece the MFMAs consume next: own DMA share landed -> barrier (everybody's share landed, everybody is done with the
    // previous piece, whose slot the next request overwrites).  Returns the LDS address of the piece; the caller requests piece
    // (this + NSLOT - 1) with issue_piece() inside its instruction stream.
    auto piece_begin = [&]() __attribute__((always_inline)) -> int {
        wait_vm<G::KW>();
        raw_barrier();
        const int base = lds0 + rd_slot * PB;
        rd_slot = rd_slot + 1 == NSLOT ? 0 : rd_slot + 1;
        return base;
    };

    // ---- fragment read offsets (the images are built by mlp_pack_host below)
    // W1 piece [slice s][row r][4 chunks]: chunk c = 2 fh + hl of (row, slice) sits at c ^ ((r >> 2) & 3)
    const int o1_hi = fr * 64 + (((2 * fh) ^ ((fr >> 2) & 3)) << 4), o1_lo = fr * 64 + (((2 * fh + 1) ^ ((fr >> 2) & 3)) << 4);
    // W2 piece [col block j][row r][8 chunks]: chunk c = 4 s + 2 fh + hl sits at c ^ ((r >> 1) & 7)
    int o2[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl) o2[s][hl] = fr * 128 + (((4 * s + 2 * fh + hl) ^ ((fr >> 1) & 7)) << 4);

    const float ws1 = p.ws1, ws2 = p.ws2;
    // biases: copied once into LDS (behind the ring); lane (fr, fh) reads its 4-column quads b[8 g + 4 fh ..] as broadcast ds_read_b128
    // (vector-memory loads would sit on the DMA stream's vmcnt, scalar loads need a per-lane select that costs 7 VALU per element)
    {
        float* bl = reinterpret_cast<float*>(smem + NSLOT * PB);
        for (int i = tid; i < 5 * C; i += 64 * MW) bl[i] = i < 4 * C ? p.b1[i] : p.b2[i - 4 * C];
        __syncthreads();
    }
    const int bias_rd = lds0 + NSLOT * PB + 16 * fh;
#define F_LDSF(addr) (*reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((size_t)(addr)))

    f16x8 Ahi[NS], Alo[NS];
    f32x16 acc2[NJ];
    f32x16 acc1[2][CH];
    f16x8 hf[2][2][2];                               // GEMM2 operand of a hidden block: [block parity][k slice][hi, lo]

    const int lda4 = p.lda * 4;
    auto load_a = [&](int tile) __attribute__((always_inline)) {
        const int m0 = tile * BMF;
        const int rows = min(BMF, p.M - m0);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.A)) + (size_t)m0 * lda4, 0, rows * lda4, 0x00020000);
        const int vo = (32 * wave + fr) * lda4 + fh * 32;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            Ahi[s] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, vo + 64 * s, 0, 0));
            Alo[s] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, vo + 64 * s + 16, 0, 0));
        }
    };
    // ---- bias + GELU + hi / lo split of the lane's 16 accumulators of a hidden block (erfc form of act_fast<ACT_GELU>, common.h), cut
    // into NGS sub-steps per quad that are dealt out between the MFMAs of the neighbouring GEMM blocks.  A sub-step is ONE operation of
    // the sequence applied to the 4 elements of the quad = 4 independent instructions (a wave alone on its SIMD stalls on every
    // dependent VALU pair, so the dependency always crosses an MFMA); the asm pins keep the optimiser from gathering the sub-steps again
    // or packing element pairs into v_pk_* ops (slow beside MFMAs).  Quad q = accumulators 4q .. 4q+3 = hidden units 8q + 4fh + e of
    // the block = elements 4 (q & 1) .. of the k-slice q >> 1 fragment of GEMM2.
    constexpr int NGS = 19;
    float gx[2][4], ga[2][4], gt[2][4], gq[2][4];    // per quad in flight (two quads of a block are interleaved)
    auto gelu_step = [&](auto K, auto QI, int q, int hb, const f32x16 (&a)[CH], f16x8 (&dst)[2][2]) __attribute__((always_inline)) {
        constexpr int k = decltype(K)::value, qi = decltype(QI)::value;
        float(&x)[4] = gx[qi];
        float(&e_)[4] = ga[qi];
        float(&t)[4] = gt[qi];
        float(&w)[4] = gq[qi];
#define G4(expr) _Pragma("unroll") for (int e = 0; e < 4; ++e) { expr; }
#define GPIN(arr) asm volatile("" : "+v"(arr[0]), "+v"(arr[1]), "+v"(arr[2]), "+v"(arr[3]))
        if constexpr (k == 0) {
            const f32x4 b4 = F_LDSF(bias_rd + (32 * hb + 8 * q) * 4);
            G4(t[e] = b4[e]);
            G4(x[e] = a[0][4 * q + e]);
            if (CH == 2) { G4(x[e] += a[CH - 1][4 * q + e]); }
            if (DBG & 16) { G4(w[e] = x[e]); }
            GPIN(x); GPIN(t);
        } else if constexpr ((DBG & 16) != 0 && k < 18) {      // ablation: no GELU / split arithmetic
        } else if constexpr (k == 1) { G4(x[e] = fmaf(x[e], ws1, t[e])); GPIN(x); }
        else if constexpr (k == 2) { G4(e_[e] = fabsf(x[e]) * 0.84932180028801904f); GPIN(e_); }
        else if constexpr (k == 3) { G4(e_[e] = -e_[e] * e_[e]); GPIN(e_); }
        else if constexpr (k == 4) { G4(e_[e] = __builtin_amdgcn_exp2f(e_[e])); GPIN(e_); }
        else if constexpr (k == 5) { G4(t[e] = fmaf(fabsf(x[e]), 0.23164188588f, 1.f)); GPIN(t); }
        else if constexpr (k == 6) { G4(t[e] = __builtin_amdgcn_rcpf(t[e])); GPIN(t); }
        else if constexpr (k == 7) { G4(w[e] = fmaf(0.5307027145f, t[e], -0.7265760135f)); GPIN(w); }
        else if constexpr (k == 8) { G4(w[e] = fmaf(w[e], t[e], 0.7107068705f)); GPIN(w); }
        else if constexpr (k == 9) { G4(w[e] = fmaf(w[e], t[e], -0.142248368f)); GPIN(w); }
        else if constexpr (k == 10) { G4(w[e] = fmaf(w[e], t[e], 0.127414796f)); GPIN(w); }
        else if constexpr (k == 11) { G4(w[e] = w[e] * t[e]); GPIN(w); }
        else if constexpr (k == 12) { G4(w[e] = w[e] * e_[e]); GPIN(w); }
        else if constexpr (k == 13) { G4(t[e] = fmaxf(x[e], 0.f)); GPIN(t); }
        else if constexpr (k == 14) { G4(x[e] = fmaf(-fabsf(x[e]), w[e], t[e])); GPIN(x); }                 // x = GELU value y
        else if constexpr (k == 15) { G4(t[e] = __builtin_amdgcn_fmed3f(x[e], -65504.f, 65504.f)); GPIN(t); }
        else if constexpr (k == 16) { G4(w[e] = (float)(f16)t[e]); GPIN(w); }                               // hi as f32 (t keeps the clamped value)
        else if constexpr (k == 17) { G4(x[e] = t[e] - w[e]); GPIN(x); }                                    // lo (of the clamped value: saturates, common.h h2_split)
        else {
            typedef __attribute__((ext_vector_type(2))) _Float16 f16x2v;
            u32x4 dh = __builtin_bit_cast(u32x4, dst[q >> 1][0]), dl = __builtin_bit_cast(u32x4, dst[q >> 1][1]);
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                f16x2v ph = {(f16)w[2 * h2], (f16)w[2 * h2 + 1]}, pl = {(f16)x[2 * h2], (f16)x[2 * h2 + 1]};
                dh[2 * (q & 1) + h2] = __builtin_bit_cast(unsigned, ph);
                dl[2 * (q & 1) + h2] = __builtin_bit_cast(unsigned, pl);
            }
            dst[q >> 1][0] = __builtin_bit_cast(f16x8, dh);
            dst[q >> 1][1] = __builtin_bit_cast(f16x8, dl);
        }
#undef G4
#undef GPIN
    };
    // sub-step `i` of the 2 * NGS that finish quads (q0, q0 + 1) of hidden block hb: the two quads alternate
    constexpr int NSUB = 2 * NGS;
    auto gelu_sub = [&](auto I, int q0, int hb, const f32x16 (&a)[CH], f16x8 (&dst)[2][2]) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        gelu_step(std::integral_constant<int, (i >> 1)>{}, std::integral_constant<int, (i & 1)>{}, q0 + (i & 1), hb, a, dst);
    };
    // A block of MFMAs as NPAIR pair-steps: two independent accumulators alternate (m(a,t0) m(b,t0) m(a,t1) m(b,t1) m(a,t2) m(b,t2) --
    // an issue slot between two MFMAs on the SAME accumulator costs ~43 cycles, on different ones ~6), the 4 fragments of the next
    // pair-step are requested after the second MFMA, one "filler" call follows every MFMA, and sched_barrier(0) pins that order: with
    // one wave per SIMD the instruction stream IS the schedule.
    f16x8 wf[2][2][2];                               // [buffer][operand of the pair][hi, lo]
    auto run_block = [&](auto NPAIR_T, auto&& frag_addr, auto&& mfma, auto&& filler) __attribute__((always_inline)) {
        constexpr int NPAIR = decltype(NPAIR_T)::value;
        auto load = [&](auto P) __attribute__((always_inline)) {
            constexpr int pp = decltype(P)::value;
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                wf[pp & 1][w][0] = F_LDS(frag_addr(2 * pp + w, 0));
                wf[pp & 1][w][1] = F_LDS(frag_addr(2 * pp + w, 1));
            }
        };
        load(std::integral_constant<int, 0>{});
        static_for<0, NPAIR>([&](auto P) __attribute__((always_inline)) {
            constexpr int pp = decltype(P)::value;
            static_for<0, 6>([&](auto Kk) __attribute__((always_inline)) {
                constexpr int k = decltype(Kk)::value;
                constexpr int w = k & 1, term = k >> 1;
                // terms: lo.hi, hi.lo, hi.hi
                if (!(DBG & 2)) mfma(2 * pp + w, term == 0 ? wf[pp & 1][w][1] : wf[pp & 1][w][0], term);
                if constexpr (k == 1 && pp + 1 < NPAIR) load(std::integral_constant<int, pp + 1>{});
                filler(std::integral_constant<int, pp * 6 + k>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    };
    // sub-steps [i0, i1) of the 32 for MFMA slot `slot` of `nslots`, the first `head` slots left free
    // GEMM1 of one hidden block: acc (2 chains: even / odd k slices) = W1[32 units][C] . a, fillers = GELU sub-steps + this piece's DMA requests
    auto block_a = [&](int base, f32x16 (&a)[CH], auto&& sub) __attribute__((always_inline)) {
        static_assert(CH == 2, "two accumulator chains");
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) a[c][r] = 0.f;
        const __attribute__((address_space(3))) char* b_hi = (const __attribute__((address_space(3))) char*)(size_t)(base + o1_hi);
        const __attribute__((address_space(3))) char* b_lo = (const __attribute__((address_space(3))) char*)(size_t)(base + o1_lo);
        constexpr int NM = 3 * NS;
        run_block(std::integral_constant<int, NS / 2>{},
                  [&](int s, int hl) __attribute__((always_inline)) { return (hl ? b_lo : b_hi) + s * 2048; },
                  [&](int s, const f16x8& w, int term) __attribute__((always_inline)) { F_MFMA(w, term == 1 ? Alo[s] : Ahi[s], a[s & 1]); },
                  [&](auto SL) __attribute__((always_inline)) {
                      constexpr int sl = decltype(SL)::value;
                      if constexpr (sl * IPW / NM != (sl + 1) * IPW / NM) issue_piece_part(std::integral_constant<int, sl * IPW / NM>{});
                      static_for<sl * NSUB / NM, (sl + 1) * NSUB / NM>([&](auto I) __attribute__((always_inline)) { sub(I); });
                  });
        issue_piece_done();
    };
    // GEMM2 of one hidden block: acc2[j] += W2'[32 j ..][32 units] . hfr; items (k slice s, column block j) in pairs
    auto block_b = [&](int base, const f16x8 (&hfr)[2][2], auto&& sub, auto HEAD, auto&& extra) __attribute__((always_inline)) {
        constexpr int NM = 6 * NJ;
        constexpr int nfill = decltype(HEAD)::value;   // the sub-steps are dealt out over the first nfill slots
        const __attribute__((address_space(3))) char* b2[2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) b2[s][hl] = (const __attribute__((address_space(3))) char*)(size_t)(base + o2[s][hl]);
        run_block(std::integral_constant<int, NJ>{},
                  [&](int it, int hl) __attribute__((always_inline)) { return b2[it / NJ][hl] + (it % NJ) * 4096; },
                  [&](int it, const f16x8& w, int term) __attribute__((always_inline)) { F_MFMA(w, term == 1 ? hfr[it / NJ][1] : hfr[it / NJ][0], acc2[it % NJ]); },
                  [&](auto SL) __attribute__((always_inline)) {
                      constexpr int sl = decltype(SL)::value;
                      if constexpr (sl * IPW / NM != (sl + 1) * IPW / NM) issue_piece_part(std::integral_constant<int, sl * IPW / NM>{});
                      if constexpr (sl < nfill) static_for<sl * NSUB / nfill, (sl + 1) * NSUB / nfill>([&](auto I) __attribute__((always_inline)) { sub(I); });
                      extra(SL);
                  });
        issue_piece_done();
    };
    auto nosub = [](auto) __attribute__((always_inline)) {};

    // ---- prologue: NSLOT - 1 pieces requested, A of the first tile
#pragma unroll
    for (int i = 0; i < NSLOT - 1; ++i) issue_piece();
    load_a(blockIdx.x);

#pragma unroll 1
    for (int t = 0; t < count; ++t) {
        const int tile = blockIdx.x + t * gridDim.x;
        const int m0 = tile * BMF;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;
        // piece order of a tile: W1_0 | W1_1 W2_0 | W1_2 W2_1 | ... | W1_{NH-1} W2_{NH-2} | W2_{NH-1}, hidden-block indices relative to h0
        // schedule: block A_h = GEMM1(h + 1) beside the second half of GELU(h); block B_h = GEMM2(h) beside the first half of GELU(h + 1)
        block_a(piece_begin(), acc1[0], nosub);
        static_for<0, NSUB>([&](auto I) __attribute__((always_inline)) { gelu_sub(I, 0, hb_of(0), acc1[0], hf[0]); });
        constexpr auto FULL = std::integral_constant<int, 6 * NJ>{};
#pragma unroll 1
        for (int h = 0; h < NH - 2; h += 2) {        // NH is even: pairs (h, h + 1), then the single block NH - 2 below
            block_a(piece_begin(), acc1[1], [&](auto I) __attribute__((always_inline)) { gelu_sub(I, 2, hb_of(h), acc1[0], hf[0]); });
            block_b(piece_begin(), hf[0], [&](auto I) __attribute__((always_inline)) { gelu_sub(I, 0, hb_of(h + 1), acc1[1], hf[1]); }, FULL, nosub);
            block_a(piece_begin(), acc1[0], [&](auto I) __attribute__((always_inline)) { gelu_sub(I, 2, hb_of(h + 1), acc1[1], hf[1]); });
            block_b(piece_begin(), hf[1], [&](auto I) __attribute__((always_inline)) { gelu_sub(I, 0, hb_of(h + 2), acc1[0], hf[0]); }, FULL, nosub);
        }
        block_a(piece_begin(), acc1[1], [&](auto I) __attribute__((always_inline)) { gelu_sub(I, 2, hb_of(NH - 2), acc1[0], hf[0]); });
        block_b(piece_begin(), hf[0], [&](auto I) __attribute__((always_inline)) { gelu_sub(I, 0, hb_of(NH - 1), acc1[1], hf[1]); }, FULL, nosub);
        // the A fragments are dead: their registers take the residual rows (accumulator layout: 16 B of row fr per (j, g)), requested
        // beside the last GEMM2
        const int rows = min(BMF, p.M - m0);
        const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res) + (size_t)m0 * p.ldr, 0, rows * p.ldr * 4, 0x00020000);
        const int vo_r = ((32 * wave + fr) * p.ldr + 4 * fh) * 4;
        f32x4 rv[NJ][4];
        // last block: the second half of GELU(NH - 1) must be complete before the k-slice-1 items (second half of the slots), so its
        // sub-steps are dealt out over the first third; the residual requests follow, one per slot
        block_b(piece_begin(), hf[1], [&](auto I) __attribute__((always_inline)) { gelu_sub(I, 2, hb_of(NH - 1), acc1[1], hf[1]); },
                std::integral_constant<int, 2 * NJ>{}, [&](auto SL) __attribute__((always_inline)) {
                    constexpr int sl = decltype(SL)::value;
                    constexpr int r0 = sl * 4 * NJ / (6 * NJ), r1 = (sl + 1) * 4 * NJ / (6 * NJ);
                    static_for<r0, r1>([&](auto R) __attribute__((always_inline)) {
                        constexpr int r = decltype(R)::value;
                        rv[r / 4][r % 4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_r, vo_r + (32 * (r / 4) + 8 * (r % 4)) * 4, 0, 0));
                    });
                });
        // ---- epilogue: out = acc2 * ws2 + b2 + res, lane (fr, fh) owns columns 32 j + 8 g + 4 fh + e of row fr
        {
            const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)m0 * p.ldo, 0, rows * p.ldo * 4, 0x00020000);
            const int vo_o = ((32 * wave + fr) * p.ldo + 4 * fh) * 4;
            const int row = m0 + 32 * wave + fr;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b4 = F_LDSF(bias_rd + (4 * C + 32 * j + 8 * g) * 4);
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(acc2[j][4 * g + e], ws2, b4[e]) + rv[j][g][e];
                    if (!(DBG & 4)) {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_o, vo_o + (32 * j + 8 * g) * 4, 0, 0);
                        if (OUTB && row < p.M) act_store4(p.outB, (size_t)row * p.ldb + 32 * j + 8 * g + 4 * fh, v[0], v[1], v[2], v[3], FMT_H2);
                    }
                }
            }
        }
        if (t + 1 < count) load_a(tile + gridDim.x);
    }
    wait_vm<0>();
}

// ================================================================================================
// 16-row variant: 8 waves per block (TWO per SIMD, 256 registers each), a wave owns 16 pixel rows and multiplies with
// v_mfma_f32_16x16x32_f16.  Same data flow (weights as a DMA-fed stream of pieces, A once per tile into fragments, GEMM1 -> registers
// -> GELU -> GEMM2), but per wave half the registers (A: C / 4, output accumulators: C / 4), so two waves share a SIMD: twice the issue
// slots per MFMA cycle and the partner's MFMAs under every VALU / LDS wait.  The 32-row kernel above is issue-bound at one wave per
// SIMD (a wave can issue one instruction per ~4 cycles; GELU + split are ~22 VALU per element).  Price: every weight fragment feeds
// half the MFMA work, i.e. twice the LDS read traffic per flop.
// Lane (pr, kg) = (lane & 15, lane >> 4) holds pixel row pr; an MFMA contracts 32 k, lane group kg supplies k = 8 kg .. 8 kg + 7.
// GEMM1 (weights first): two 16-unit blocks a / b of a hidden 32-block leave the lane with units 4 kg + r (a) and 16 + 4 kg + r (b);
// these 8 values are the lane's GEMM2 operand for k positions 8 kg + i when W2 is packed with position 8 kg + i <-> unit
// (i < 4 ? 4 kg + i : 16 + 4 kg + i - 4).
// ================================================================================================
namespace {
constexpr int MW16 = 8;
template <int C>
struct Geo16 {
    static constexpr int NS = C / 32;                // k slices (32 k) of the A operand
    static constexpr int NH = C / 8;                 // hidden blocks of 32 units
    static constexpr int NJ = C / 16;                // output column blocks of 16
    static constexpr int PB = 128 * C;
    static constexpr int IPW = C / 64;               // 1-KiB DMA instructions per wave and piece
    static constexpr int BIAS = 20 * C;
    static constexpr int NSLOT = ((163840 - BIAS) / PB) < 6 ? ((163840 - BIAS) / PB) : 6;
    static constexpr int LDS = NSLOT * PB + BIAS;
    static constexpr int NP = 2 * NH;
    static constexpr int KW = (NSLOT - 2) * IPW;
    static_assert(C % 64 == 0 && NSLOT >= 3 && KW <= 60, "geometry");
};
#define F_MFMA16(w, a, c) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, a, c, 0, 0, 0)
}  // namespace

template <int C, bool OUTB, int DBG>
__global__ __launch_bounds__(64 * MW16, 2) void mlp_fused16_kernel(MlpArgs p) {
    using G = Geo16<C>;
    constexpr int NS = G::NS, NH = G::NH, NJ = G::NJ, PB = G::PB, IPW = G::IPW, NSLOT = G::NSLOT, NP = G::NP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pr = lane & 15, kg = lane >> 4;

    const int ntiles = (p.M + BMF - 1) / BMF;
    const int count = ((int)blockIdx.x < ntiles) ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    if (count == 0) return;
    const int total = count * NP;

    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.blob), 0, NP * PB, 0x00020000);
    const int vo_w = lane * 16 + wave * 1024;
    const int lds0 = (int)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int h0 = (p.dbg & 8) ? 0 : (int)(((blockIdx.x >> 3) * 13u + (blockIdx.x & 7) * 5u) % NH);
    const int q_first = 2 * h0, q_last = 2 * h0 + 1;
    auto hb_of = [&](int step) __attribute__((always_inline)) { const int v = h0 + step; return v >= NH ? v - NH : v; };
    int p_issue = 0, src_q = q_first, wr_slot = 0;
    auto issue_piece_part = [&](auto II) __attribute__((always_inline)) {
        constexpr int i = decltype(II)::value;
        const int so = (p_issue < total && !(DBG & 1)) ? src_q * PB : 0x40000000;
        char* dst = smem + wr_slot * PB + wave * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(dst + i * 8192), 16, vo_w, so + i * 8192, 0, 0);
    };
    auto issue_piece_done = [&]() __attribute__((always_inline)) {
        ++p_issue;
        if (src_q == q_last) src_q = q_first;
        else {
            int n = src_q == q_first ? src_q + 2 : src_q + 1;
            n = n >= NP ? n - NP : n;
            src_q = n == q_first ? q_last : n;
        }
        wr_slot = wr_slot + 1 == NSLOT ? 0 : wr_slot + 1;
    };
    auto issue_piece = [&]() __attribute__((always_inline)) {
        static_for<0, IPW>([&](auto II) __attribute__((always_inline)) { issue_piece_part(II); });
        issue_piece_done();
    };
    int rd_slot = 0;
    auto piece_begin = [&]() __attribute__((always_inline)) -> int {
        wait_vm<G::KW>();
        raw_barrier();
        const int base = lds0 + rd_slot * PB;
        rd_slot = rd_slot + 1 == NSLOT ? 0 : rd_slot + 1;
        return base;
    };
    // fragment images (mlp_pack16_host): [..][kg (4)][hi, lo][row (16)] x 16 B: a 16-lane group of a ds_read_b128 covers 256 contiguous
    // bytes (or two 128-byte halves of different kg blocks that fall on disjoint banks): conflict-free without a swizzle
    const int o_frag = kg * 512 + pr * 16;

    const float ws1 = p.ws1, ws2 = p.ws2;
    {
        float* bl = reinterpret_cast<float*>(smem + NSLOT * PB);
        for (int i = tid; i < 5 * C; i += 64 * MW16) bl[i] = i < 4 * C ? p.b1[i] : p.b2[i - 4 * C];
        __syncthreads();
    }
    const int bias_rd = lds0 + NSLOT * PB + 16 * kg;       // lane's 4 consecutive units / channels 4 kg .. 4 kg + 3 of a 16-block

    f16x8 Ahi[NS], Alo[NS];
    f32x4 acc2[NJ];
    f32x4 acc1[2][2];                                // [hidden-block parity][unit block a / b]
    f16x8 hf[2][2];                                  // GEMM2 operand of a hidden block: [parity][hi, lo]

    const int lda4 = p.lda * 4;
    auto load_a = [&](int tile) __attribute__((always_inline)) {
        const int m0 = tile * BMF;
        const int rows = min(BMF, p.M - m0);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.A)) + (size_t)m0 * lda4, 0, rows * lda4, 0x00020000);
        const int vo = (16 * wave + pr) * lda4 + kg * 32;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            Ahi[s] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, vo + 128 * s, 0, 0));
            Alo[s] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, vo + 128 * s + 16, 0, 0));
        }
    };
    // GELU + split: quad ub = the lane's 4 accumulators of unit block ub = elements 4 ub .. 4 ub + 3 of the GEMM2 fragment
    constexpr int NGS = 19;
    float gx[2][4], ga[2][4], gt[2][4], gq[2][4];
    auto gelu_step = [&](auto K, auto UB, int hb, const f32x4 (&a)[2], f16x8 (&dst)[2]) __attribute__((always_inline)) {
        constexpr int k = decltype(K)::value, ub = decltype(UB)::value;
        float(&x)[4] = gx[ub];
        float(&e_)[4] = ga[ub];
        float(&t)[4] = gt[ub];
        float(&w)[4] = gq[ub];
#define G4(expr) _Pragma("unroll") for (int e = 0; e < 4; ++e) { expr; }
#define GPIN(arr) asm volatile("" : "+v"(arr[0]), "+v"(arr[1]), "+v"(arr[2]), "+v"(arr[3]))
        if constexpr (k == 0) {
            const f32x4 b4 = F_LDSF(bias_rd + (32 * hb + 16 * ub) * 4);
            G4(t[e] = b4[e]);
            G4(x[e] = a[ub][e]);
            if (DBG & 16) { G4(w[e] = x[e]); }
            GPIN(x); GPIN(t);
        } else if constexpr ((DBG & 16) != 0 && k < 18) {
        } else if constexpr (k == 1) { G4(x[e] = fmaf(x[e], ws1, t[e])); GPIN(x); }
        else if constexpr (k == 2) { G4(e_[e] = fabsf(x[e]) * 0.84932180028801904f); GPIN(e_); }
        else if constexpr (k == 3) { G4(e_[e] = -e_[e] * e_[e]); GPIN(e_); }
        else if constexpr (k == 4) { G4(e_[e] = __builtin_amdgcn_exp2f(e_[e])); GPIN(e_); }
        else if constexpr (k == 5) { G4(t[e] = fmaf(fabsf(x[e]), 0.23164188588f, 1.f)); GPIN(t); }
        else if constexpr (k == 6) { G4(t[e] = __builtin_amdgcn_rcpf(t[e])); GPIN(t); }
        else if constexpr (k == 7) { G4(w[e] = fmaf(0.5307027145f, t[e], -0.7265760135f)); GPIN(w); }
        else if constexpr (k == 8) { G4(w[e] = fmaf(w[e], t[e], 0.7107068705f)); GPIN(w); }
        else if constexpr (k == 9) { G4(w[e] = fmaf(w[e], t[e], -0.142248368f)); GPIN(w); }
        else if constexpr (k == 10) { G4(w[e] = fmaf(w[e], t[e], 0.127414796f)); GPIN(w); }
        else if constexpr (k == 11) { G4(w[e] = w[e] * t[e]); GPIN(w); }
        else if constexpr (k == 12) { G4(w[e] = w[e] * e_[e]); GPIN(w); }
        else if constexpr (k == 13) { G4(t[e] = fmaxf(x[e], 0.f)); GPIN(t); }
        else if constexpr (k == 14) { G4(x[e] = fmaf(-fabsf(x[e]), w[e], t[e])); GPIN(x); }
        else if constexpr (k == 15) { G4(t[e] = __builtin_amdgcn_fmed3f(x[e], -65504.f, 65504.f)); GPIN(t); }
        else if constexpr (k == 16) { G4(w[e] = (float)(f16)t[e]); GPIN(w); }
        else if constexpr (k == 17) { G4(x[e] = t[e] - w[e]); GPIN(x); }
        else {
            typedef __attribute__((ext_vector_type(2))) _Float16 f16x2v;
            u32x4 dh = __builtin_bit_cast(u32x4, dst[0]), dl = __builtin_bit_cast(u32x4, dst[1]);
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                f16x2v ph = {(f16)w[2 * h2], (f16)w[2 * h2 + 1]}, pl = {(f16)x[2 * h2], (f16)x[2 * h2 + 1]};
                dh[2 * ub + h2] = __builtin_bit_cast(unsigned, ph);
                dl[2 * ub + h2] = __builtin_bit_cast(unsigned, pl);
            }
            dst[0] = __builtin_bit_cast(f16x8, dh);
            dst[1] = __builtin_bit_cast(f16x8, dl);
        }
#undef G4
#undef GPIN
    };
    f16x8 wf[2][2][2];
    auto run_block = [&](auto NPAIR_T, auto&& frag_addr, auto&& mfma, auto&& filler) __attribute__((always_inline)) {
        constexpr int NPAIR = decltype(NPAIR_T)::value;
        auto load = [&](auto P) __attribute__((always_inline)) {
            constexpr int pp = decltype(P)::value;
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                wf[pp & 1][w][0] = F_LDS(frag_addr(2 * pp + w, 0));
                wf[pp & 1][w][1] = F_LDS(frag_addr(2 * pp + w, 1));
            }
        };
        load(std::integral_constant<int, 0>{});
        static_for<0, NPAIR>([&](auto P) __attribute__((always_inline)) {
            constexpr int pp = decltype(P)::value;
            static_for<0, 6>([&](auto Kk) __attribute__((always_inline)) {
                constexpr int k = decltype(Kk)::value;
                constexpr int w = k & 1, term = k >> 1;
                if (!(DBG & 2)) mfma(2 * pp + w, term == 0 ? wf[pp & 1][w][1] : wf[pp & 1][w][0], term);
                if constexpr (k == 1 && pp + 1 < NPAIR) load(std::integral_constant<int, pp + 1>{});
                filler(std::integral_constant<int, pp * 6 + k>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    };
    // block A: GEMM1 of a hidden block, items = (slice s, unit block ub) = 2 s + ub; fillers: GELU quad b (ub = 1) of the previous block
    auto block_a = [&](int base, f32x4 (&a)[2], auto&& sub) __attribute__((always_inline)) {
        a[0] = f32x4{0.f, 0.f, 0.f, 0.f};
        a[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        const __attribute__((address_space(3))) char* b0 = (const __attribute__((address_space(3))) char*)(size_t)(base + o_frag);
        constexpr int NM = 6 * NS;
        run_block(std::integral_constant<int, NS>{},
                  [&](int it, int hl) __attribute__((always_inline)) { return b0 + (it & 1) * (NS * 2048) + (it >> 1) * 2048 + hl * 256; },
                  [&](int it, const f16x8& w, int term) __attribute__((always_inline)) { F_MFMA16(w, term == 1 ? Alo[it >> 1] : Ahi[it >> 1], a[it & 1]); },
                  [&](auto SL) __attribute__((always_inline)) {
                      constexpr int sl = decltype(SL)::value;
                      if constexpr (sl * IPW / NM != (sl + 1) * IPW / NM) issue_piece_part(std::integral_constant<int, sl * IPW / NM>{});
                      static_for<sl * NGS / NM, (sl + 1) * NGS / NM>([&](auto I) __attribute__((always_inline)) { sub(I); });
                  });
        issue_piece_done();
    };
    // block B: GEMM2 of a hidden block, items = output column blocks j; fillers: GELU quad a (ub = 0) of the next block
    auto block_b = [&](int base, const f16x8 (&hfr)[2], auto&& sub, auto HEAD, auto&& extra) __attribute__((always_inline)) {
        constexpr int NM = 3 * NJ;
        constexpr int nfill = decltype(HEAD)::value;
        const __attribute__((address_space(3))) char* b0 = (const __attribute__((address_space(3))) char*)(size_t)(base + o_frag);
        run_block(std::integral_constant<int, NJ / 2>{},
                  [&](int it, int hl) __attribute__((always_inline)) { return b0 + it * 2048 + hl * 256; },
                  [&](int it, const f16x8& w, int term) __attribute__((always_inline)) { F_MFMA16(w, term == 1 ? hfr[1] : hfr[0], acc2[it]); },
                  [&](auto SL) __attribute__((always_inline)) {
                      constexpr int sl = decltype(SL)::value;
                      if constexpr (sl * IPW / NM != (sl + 1) * IPW / NM) issue_piece_part(std::integral_constant<int, sl * IPW / NM>{});
                      if constexpr (sl < nfill) static_for<sl * NGS / nfill, (sl + 1) * NGS / nfill>([&](auto I) __attribute__((always_inline)) { sub(I); });
                      extra(SL);
                  });
        issue_piece_done();
    };
    auto nosub = [](auto) __attribute__((always_inline)) {};
    constexpr auto QA = std::integral_constant<int, 0>{};
    constexpr auto QB_ = std::integral_constant<int, 1>{};

#pragma unroll
    for (int i = 0; i < NSLOT - 1; ++i) issue_piece();
    load_a(blockIdx.x);

#pragma unroll 1
    for (int t = 0; t < count; ++t) {
        const int tile = blockIdx.x + t * gridDim.x;
        const int m0 = tile * BMF;
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc2[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // schedule: block A_h = GEMM1(h + 1) beside GELU quad b of h; block B_h = GEMM2(h) beside GELU quad a of h + 1
        block_a(piece_begin(), acc1[0], nosub);
        static_for<0, NGS>([&](auto I) __attribute__((always_inline)) { gelu_step(I, QA, hb_of(0), acc1[0], hf[0]); });
        constexpr auto FULL = std::integral_constant<int, 3 * NJ>{};
#pragma unroll 1
        for (int h = 0; h < NH - 2; h += 2) {
            block_a(piece_begin(), acc1[1], [&](auto I) __attribute__((always_inline)) { gelu_step(I, QB_, hb_of(h), acc1[0], hf[0]); });
            block_b(piece_begin(), hf[0], [&](auto I) __attribute__((always_inline)) { gelu_step(I, QA, hb_of(h + 1), acc1[1], hf[1]); }, FULL, nosub);
            block_a(piece_begin(), acc1[0], [&](auto I) __attribute__((always_inline)) { gelu_step(I, QB_, hb_of(h + 1), acc1[1], hf[1]); });
            block_b(piece_begin(), hf[1], [&](auto I) __attribute__((always_inline)) { gelu_step(I, QA, hb_of(h + 2), acc1[0], hf[0]); }, FULL, nosub);
        }
        block_a(piece_begin(), acc1[1], [&](auto I) __attribute__((always_inline)) { gelu_step(I, QB_, hb_of(NH - 2), acc1[0], hf[0]); });
        block_b(piece_begin(), hf[0], [&](auto I) __attribute__((always_inline)) { gelu_step(I, QA, hb_of(NH - 1), acc1[1], hf[1]); }, FULL, nosub);
        // last hidden block: its quad b has no GEMM1 block to hide in -> done before its GEMM2; the residual rows are requested beside it
        static_for<0, NGS>([&](auto I) __attribute__((always_inline)) { gelu_step(I, QB_, hb_of(NH - 1), acc1[1], hf[1]); });
        const int rows = min(BMF, p.M - m0);
        const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res) + (size_t)m0 * p.ldr, 0, rows * p.ldr * 4, 0x00020000);
        const int vo_r = ((16 * wave + pr) * p.ldr + 4 * kg) * 4;
        f32x4 rv[NJ];
        block_b(piece_begin(), hf[1], nosub, std::integral_constant<int, 0>{}, [&](auto SL) __attribute__((always_inline)) {
            constexpr int sl = decltype(SL)::value;
            constexpr int r0 = sl * NJ / (3 * NJ), r1 = (sl + 1) * NJ / (3 * NJ);
            static_for<r0, r1>([&](auto R) __attribute__((always_inline)) {
                constexpr int r = decltype(R)::value;
                rv[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_r, vo_r + 64 * r, 0, 0));
            });
        });
        {
            const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)m0 * p.ldo, 0, rows * p.ldo * 4, 0x00020000);
            const int vo_o = ((16 * wave + pr) * p.ldo + 4 * kg) * 4;
            const int row = m0 + 16 * wave + pr;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const f32x4 b4 = F_LDSF(bias_rd + (4 * C + 16 * j) * 4);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(acc2[j][e], ws2, b4[e]) + rv[j][e];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_o, vo_o + 64 * j, 0, 0);
                if (OUTB && row < p.M) act_store4(p.outB, (size_t)row * p.ldb + 16 * j + 4 * kg, v[0], v[1], v[2], v[3], FMT_H2);
            }
        }
        if (t + 1 < count) load_a(tile + gridDim.x);
    }
    wait_vm<0>();
}

// ------------------------------------------------------------------------------------------------
// host: weight stream in consumption order, every piece in its LDS image
// ------------------------------------------------------------------------------------------------
// w1 [4C][C], w2 [C][4C] fp32 (nn.Linear layouts of pwconv1 / pwconv2), gamma [C] = layer scale folded into the rows of w2 (may be
// NULL).  out: mlp_blob_bytes(C) bytes.  *ws1 / *ws2 = the factors the accumulators are multiplied with (1 / power-of-two scale).
size_t mlp_blob_bytes(int C) { return (size_t)32 * C * C; }
// C = 384 would need 192 (A) + 192 (output accumulators) registers before anything else: it spills, and a 64-row tile that halves them
// is DMA-bound (the whole 4.7 MB stream per 64 rows).  Stage 1 of the large model stays on the unfused pair.
bool mlp_fused_supported(int C) { return C == 96 || C == 192 || C == 256; }

void mlp_pack_host(const float* w1, const float* w2, const float* gamma, int C, uint16_t* out, float* ws1, float* ws2) {
    const int H4 = 4 * C, NH = C / 8, NS = C / 16, NJ = C / 32, PB = 128 * C;
    float m1 = 0.f, m2 = 0.f;
    for (size_t i = 0; i < (size_t)H4 * C; ++i) m1 = std::max(m1, fabsf(w1[i]));
    for (int n = 0; n < C; ++n)
        for (int k = 0; k < H4; ++k) m2 = std::max(m2, fabsf((gamma ? gamma[n] : 1.f) * w2[(size_t)n * H4 + k]));
    const float s1 = h2_weight_scale(m1), s2 = h2_weight_scale(m2);
    *ws1 = 1.f / s1;
    *ws2 = 1.f / s2;
    auto put8 = [](uint16_t* dst, const float* v8, bool lo) {
        for (int i = 0; i < 8; ++i) {
            const uint16_t hi = f32_to_f16_host(v8[i]);
            dst[i] = lo ? f32_to_f16_host(v8[i] - f16_to_f32_host(hi)) : hi;
        }
    };
    auto piece_w1 = [&](int h, uint16_t* dst) {      // [slice][row][4 chunks of 16 B]
        for (int s = 0; s < NS; ++s)
            for (int r = 0; r < 32; ++r)
                for (int pc = 0; pc < 4; ++pc) {
                    const int c = pc ^ ((r >> 2) & 3), fh = c >> 1, hl = c & 1;
                    float v8[8];
                    for (int i = 0; i < 8; ++i) v8[i] = w1[(size_t)(32 * h + r) * C + 16 * s + 8 * fh + i] * s1;
                    put8(dst + ((size_t)(s * 32 + r) * 4 + pc) * 8, v8, hl);
                }
    };
    auto piece_w2 = [&](int h, uint16_t* dst) {      // [col block][row][8 chunks of 16 B], k order inside 16 hidden units: 0-3 8-11 | 4-7 12-15
        for (int j = 0; j < NJ; ++j)
            for (int r = 0; r < 32; ++r)
                for (int pc = 0; pc < 8; ++pc) {
                    const int c = pc ^ ((r >> 1) & 7), s = c >> 2, fh = (c >> 1) & 1, hl = c & 1;
                    const int n = 32 * j + r;
                    const float gsc = (gamma ? gamma[n] : 1.f) * s2;
                    float v8[8];
                    for (int i = 0; i < 8; ++i) {
                        const int u = 32 * h + 16 * s + (i < 4 ? 4 * fh + i : 4 + 4 * fh + i);
                        v8[i] = gsc * w2[(size_t)n * H4 + u];
                    }
                    put8(dst + ((size_t)(j * 32 + r) * 8 + pc) * 8, v8, hl);
                }
    };
    const size_t pe = (size_t)PB / 2;                // uint16 per piece
    for (int i = 0; i < NH; ++i) {                   // blob position 2 i: W1_i, 2 i + 1: W2_{i-1 mod NH} (the kernel's cyclic consumption order)
        piece_w1(i, out + (size_t)(2 * i) * pe);
        piece_w2((i + NH - 1) % NH, out + (size_t)(2 * i + 1) * pe);
    }
}

void mlp_pack16_host(const float* w1, const float* w2, const float* gamma, int C, uint16_t* out, float* ws1, float* ws2) {
    const int H4 = 4 * C, NH = C / 8, NS = C / 32, NJ = C / 16, PB = 128 * C;
    float m1 = 0.f, m2 = 0.f;
    for (size_t i = 0; i < (size_t)H4 * C; ++i) m1 = std::max(m1, fabsf(w1[i]));
    for (int n = 0; n < C; ++n)
        for (int k = 0; k < H4; ++k) m2 = std::max(m2, fabsf((gamma ? gamma[n] : 1.f) * w2[(size_t)n * H4 + k]));
    const float s1 = h2_weight_scale(m1), s2 = h2_weight_scale(m2);
    *ws1 = 1.f / s1;
    *ws2 = 1.f / s2;
    auto put8 = [](uint16_t* dst, const float* v8, bool lo) {
        for (int i = 0; i < 8; ++i) {
            const uint16_t hi = f32_to_f16_host(v8[i]);
            dst[i] = lo ? f32_to_f16_host(v8[i] - f16_to_f32_host(hi)) : hi;
        }
    };
    auto piece_w1 = [&](int h, uint16_t* dst) {      // [unit block ub][slice s][kg][hi, lo][row 16] x 16 B
        for (int ub = 0; ub < 2; ++ub)
            for (int s = 0; s < NS; ++s)
                for (int kg = 0; kg < 4; ++kg)
                    for (int hl = 0; hl < 2; ++hl)
                        for (int r = 0; r < 16; ++r) {
                            float v8[8];
                            for (int i = 0; i < 8; ++i) v8[i] = w1[(size_t)(32 * h + 16 * ub + r) * C + 32 * s + 8 * kg + i] * s1;
                            put8(dst + ((((size_t)(ub * NS + s) * 4 + kg) * 2 + hl) * 16 + r) * 8, v8, hl);
                        }
    };
    auto piece_w2 = [&](int h, uint16_t* dst) {      // [column block j][kg][hi, lo][row 16] x 16 B; k position 8 kg + i <-> unit (i < 4 ? 4 kg + i : 16 + 4 kg + i - 4)
        for (int j = 0; j < NJ; ++j)
            for (int kg = 0; kg < 4; ++kg)
                for (int hl = 0; hl < 2; ++hl)
                    for (int r = 0; r < 16; ++r) {
                        const int n = 16 * j + r;
                        const float gsc = (gamma ? gamma[n] : 1.f) * s2;
                        float v8[8];
                        for (int i = 0; i < 8; ++i) {
                            const int u = 32 * h + (i < 4 ? 4 * kg + i : 16 + 4 * kg + i - 4);
                            v8[i] = gsc * w2[(size_t)n * H4 + u];
                        }
                        put8(dst + ((((size_t)j * 4 + kg) * 2 + hl) * 16 + r) * 8, v8, hl);
                    }
    };
    const size_t pe = (size_t)PB / 2;
    for (int i = 0; i < NH; ++i) {
        piece_w1(i, out + (size_t)(2 * i) * pe);
        piece_w2((i + NH - 1) % NH, out + (size_t)(2 * i + 1) * pe);
    }
}
bool mlp_fused16_supported(int C) { return C == 192 || C == 256; }

template <int C, bool OUTB, int DBG>
static int launch_mlp16_k(const MlpArgs& a, int grid, hipStream_t s) {
    constexpr int lds = Geo16<C>::LDS;
    static DevOnce attr_once;
    UNI_LDS_OPTIN(attr_once, "mlp_fused16", lds, reinterpret_cast<const void*>(&mlp_fused16_kernel<C, OUTB, DBG>));
    hipLaunchKernelGGL((mlp_fused16_kernel<C, OUTB, DBG>), dim3(grid), dim3(64 * MW16), lds, s, a);
    return 0;
}

template <int C, int CH, bool OUTB, int DBG>
static int launch_mlp_k(const MlpArgs& a, int grid, hipStream_t s) {
    constexpr int lds = Geo<C>::LDS;
    static DevOnce attr_once;
    UNI_LDS_OPTIN(attr_once, "mlp_fused", lds, reinterpret_cast<const void*>(&mlp_fused_kernel<C, CH, OUTB, DBG>));
    hipLaunchKernelGGL((mlp_fused_kernel<C, CH, OUTB, DBG>), dim3(grid), dim3(64 * MW), lds, s, a);
    return 0;
}
template <int C, int CH>
static int launch_mlp_inst(const MlpArgs& a, int grid, hipStream_t s) {
    if (a.dbg && C == 192 && !a.outB) {     // ablations for tools/mlp_bench.py, C = 192 only
        switch (a.dbg) {
            case 1: return launch_mlp_k<192, 2, false, 1>(a, grid, s);
            case 2: return launch_mlp_k<192, 2, false, 2>(a, grid, s);
            case 3: return launch_mlp_k<192, 2, false, 3>(a, grid, s);
            case 4: return launch_mlp_k<192, 2, false, 4>(a, grid, s);
            case 16: return launch_mlp_k<192, 2, false, 16>(a, grid, s);
            case 17: return launch_mlp_k<192, 2, false, 17>(a, grid, s);
            default: break;
        }
    }
    return a.outB ? launch_mlp_k<C, CH, true, 0>(a, grid, s) : launch_mlp_k<C, CH, false, 0>(a, grid, s);
}

int launch_mlp_fused(const MlpArgs& a, hipStream_t s) {
    UNI_REQUIRE(mlp_fused_supported(a.C), "mlp_fused: C=%d unsupported", a.C);
    UNI_REQUIRE(a.A && a.blob && a.b1 && a.b2 && a.res && a.out && a.M > 0, "mlp_fused: NULL argument");
    UNI_REQUIRE(a.lda % 8 == 0 && a.ldr % 4 == 0 && a.ldo % 4 == 0 && (!a.outB || a.ldb % 8 == 0), "mlp_fused: lda=%d ldr=%d ldo=%d ldb=%d", a.lda, a.ldr, a.ldo, a.ldb);
    UNI_REQUIRE(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.blob & 15) == 0 && ((uintptr_t)a.res & 15) == 0 && ((uintptr_t)a.out & 15) == 0 &&
                ((uintptr_t)a.b1 & 3) == 0 && ((uintptr_t)a.b2 & 3) == 0, "mlp_fused: unaligned pointer");
    UNI_REQUIRE((long)BMF * a.lda * 4 < (1L << 31) && (long)BMF * a.ldo * 4 < (1L << 31), "mlp_fused: row stride too large");
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
        ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int ntiles = cdiv(a.M, BMF);
    const int grid = ntiles < ncu ? ntiles : ncu;
    if (a.layout == 1) {      // 16-row waves, two per SIMD (blob of mlp_pack16_host)
        UNI_REQUIRE(mlp_fused16_supported(a.C), "mlp_fused: the 16-row variant supports C = 192 / 256 (got %d)", a.C);
        const int d = a.dbg & ~8;
        if (a.C == 192) {
            if (a.outB) return launch_mlp16_k<192, true, 0>(a, grid, s);
            if (d == 1) return launch_mlp16_k<192, false, 1>(a, grid, s);
            if (d == 16) return launch_mlp16_k<192, false, 16>(a, grid, s);
            if (d == 17) return launch_mlp16_k<192, false, 17>(a, grid, s);
            return launch_mlp16_k<192, false, 0>(a, grid, s);
        }
        return a.outB ? launch_mlp16_k<256, true, 0>(a, grid, s) : launch_mlp16_k<256, false, 0>(a, grid, s);
    }
    switch (a.C) {
        case 96: return launch_mlp_inst<96, 2>(a, grid, s);
        case 192: return launch_mlp_inst<192, 2>(a, grid, s);
        default: return launch_mlp_inst<256, 2>(a, grid, s);
    }
}
