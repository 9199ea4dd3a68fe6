// Shared epilogue of the MFMA GEMM kernels (gemm.hip: bf16 / exact-fp32 operands, gemm_h2.hip: split-f16 operands).
// The accumulator layout is the same for every operand format (C/D of v_mfma_f32_32x32x*), so bias / activation /
// residual / fp32 + operand-format outputs / GroupNorm statistics are written once here.
#pragma once
#include "kernels.h"
#include <type_traits>

// ---- LDS-staged stores.  The MFMA leaves each lane with 4-channel quads of ONE pixel row, so direct stores touch
// 64 different cache lines per wave instruction (16 B each) and the store phase runs at ~2-3 TB/s with the MFMA pipe
// idle (measured: 40-80 % on top of the K loop for the ConvNeXt MLP shapes).  Instead every wave transposes its
// 32 x (32 TN) accumulator slab through a private fp32 LDS tile (XOR-swizzled float4 chunks, conflict-free both ways;
// the operand buffers are dead by now) and writes whole rows: a wave instruction covers 8 full 128-B lines (bf16) or
// 4 x 256 B (fp32); residual reads are coalesced the same way.  Only wave-local ordering is needed (in-order LDS).
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int WM, int WN, int TM, int TN, bool STATS>
__device__ __forceinline__ void gemm_store_staged(const GemmArgs& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn,
                                                  int lane, char* smem) {
    constexpr int CW = 32 * TN;            // wave tile width in channels
    constexpr int CPRo = CW / 4;           // float4 chunks per staged row (8 or 16)
    float* st = reinterpret_cast<float*>(smem) + (wm * WN + wn) * (32 * CW);
    const int fr = lane & 31, fh = lane >> 5;
    const int nw0 = n0 + wn * CW;
    __syncthreads();                       // all waves are done with the last operand tile
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        // phase A: bias + activation in the MFMA layout, float4 chunks to LDS [row fr][chunk ^ (fr & 7)].  The activation
        // kind is dispatched ONCE per slab (uniform branch) so the 32-element loops are branch-free; the column window
        // (act_col0) is a select.
        auto phase_a = [&](auto act_tag) __attribute__((always_inline)) {
            constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = nw0 + j * 32 + 8 * g + 4 * fh;
                    f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + (col < p.N ? col : 0));
                    if (ACT != ACT_NONE) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float a = act_fast<ACT>(v[e]);
                            v[e] = (col + e >= p.act_col0) ? a : v[e];
                        }
                    }
                    const int c = j * 8 + 2 * g + fh;
                    *reinterpret_cast<f32x4*>(st + fr * CW + ((c ^ (fr & 7)) << 2)) = v;
                }
            }
        };
        if (STATS) {   // GroupNorm-statistics GEMMs never carry an activation (it follows the normalisation; launch_gemm checks)
            phase_a(std::integral_constant<int, ACT_NONE>{});
        } else switch (p.act) {
            case ACT_GELU: phase_a(std::integral_constant<int, ACT_GELU>{}); break;
            case ACT_RELU: phase_a(std::integral_constant<int, ACT_RELU>{}); break;
            case ACT_SILU: phase_a(std::integral_constant<int, ACT_SILU>{}); break;
            case ACT_SIGMOID: phase_a(std::integral_constant<int, ACT_SIGMOID>{}); break;
            default: phase_a(std::integral_constant<int, ACT_NONE>{}); break;
        }
        wave_lds_fence();
        // phase B: whole rows out
        const int rbase = m0 + wm * 32 * TM + i * 32;
        if (!p.outF && !p.res) {
            constexpr int Q = CW / 8, RPI = 64 / Q;        // 8-channel (16-B bf16) pieces per row, rows per instruction
            constexpr bool EXACT = 64 % Q == 0;            // TN = 3 (96-wide wave tiles): 60 of the 64 lanes carry a piece, 7 instructions
            const int q = lane % Q, rr = lane / Q;
            const int col = nw0 + 8 * q;
#pragma unroll
            for (int t = 0; t < (32 + RPI - 1) / RPI; ++t) {
                const int r_ = t * RPI + rr;
                const bool lane_on = EXACT || (rr < RPI && r_ < 32);
                const int r = lane_on ? r_ : 0, row = rbase + r;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(st + r * CW + (((2 * q) ^ (r & 7)) << 2));
                const f32x4 hi = *reinterpret_cast<const f32x4*>(st + r * CW + (((2 * q + 1) ^ (r & 7)) << 2));
                if (lane_on && row < p.M && col < p.N && !(p.dbg & 128)) {
                    if (p.b32 == FMT_H2) {      // split-f16 operand buffer: [8 x hi][8 x lo] per 8 channels
                        const float v8[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        act_store8(p.outB, (size_t)row * p.ldb + col, v8, FMT_H2);
                    } else {
                        bf16x2 w0 = {(bf16)lo[0], (bf16)lo[1]}, w1 = {(bf16)lo[2], (bf16)lo[3]};
                        bf16x2 w2 = {(bf16)hi[0], (bf16)hi[1]}, w3 = {(bf16)hi[2], (bf16)hi[3]};
                        u32x4 o4 = {__builtin_bit_cast(unsigned, w0), __builtin_bit_cast(unsigned, w1),
                                    __builtin_bit_cast(unsigned, w2), __builtin_bit_cast(unsigned, w3)};
                        *reinterpret_cast<u32x4*>(p.outB + (size_t)row * p.ldb + col) = o4;
                    }
                }
            }
        } else {
            constexpr int RPI = 64 / CPRo;
            constexpr bool EXACT = 64 % CPRo == 0;         // TN = 3: 48 of the 64 lanes carry a float4 (a lane past them must not repeat a row: res may alias outF)
            const int c = lane % CPRo, rr = lane / CPRo;
            const int col = nw0 + 4 * c;
#pragma unroll
            for (int t = 0; t < (32 + RPI - 1) / RPI; ++t) {
                const int r_ = t * RPI + rr;
                const bool lane_on = EXACT || (rr < RPI && r_ < 32);
                const int r = lane_on ? r_ : 0, row = rbase + r;
                f32x4 v = *reinterpret_cast<const f32x4*>(st + r * CW + ((c ^ (r & 7)) << 2));
                if (lane_on && row < p.M && col < p.N && !(p.dbg & 128)) {
                    if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + (size_t)row * p.ldr + col);
                    if (p.outF) {
                        const int orow = p.out_hw ? (row / p.out_hw) * p.out_stride + p.out_off + row % p.out_hw : row;
                        *reinterpret_cast<f32x4*>(p.outF + (size_t)orow * p.ldf + col) = v;
                    }
                    if (p.outB) {
                        if (p.b32 == FMT_H2) {
                            act_store4(p.outB, (size_t)row * p.ldb + col, v[0], v[1], v[2], v[3], FMT_H2);
                        } else {
                            bf16x2 w0 = {(bf16)v[0], (bf16)v[1]}, w1 = {(bf16)v[2], (bf16)v[3]};
                            u32x2 o2 = {__builtin_bit_cast(unsigned, w0), __builtin_bit_cast(unsigned, w1)};
                            *reinterpret_cast<u32x2*>(p.outB + (size_t)row * p.ldb + col) = o2;
                        }
                    }
                }
            }
        }
        wave_lds_fence();
    }
}

// ---- GroupNorm statistics of (acc + bias), PER SAMPLE (GemmArgs::stats): block-wide (uses __syncthreads), `red_lds` = (WM * BN * 2 + 128)
// floats of LDS nobody else touches meanwhile.  acc must already carry the weight scale.
template <int WM, int WN, int TM, int TN>
__device__ __forceinline__ void gemm_stats(const GemmArgs& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int lane, int tid,
                                           float* red_lds) {
    constexpr int BN = 32 * TN * WN, BMt = 32 * TM * WM;
    const int fr = lane & 31, fh = lane >> 5;
    {
        // GroupNorm statistics of (acc + bias), PER SAMPLE: a block tile may straddle samples of a batch, so the
        // reduction runs once per sample present in the tile (one iteration except at sample boundaries).
        // Pixel lanes are reduced with a reduce-scatter butterfly: after offsets 16,8,4,2 lane l keeps register index
        // r = (l>>1)&15 (bit4->r3, bit3->r2, bit2->r1, bit1->r0); offset 1 adds the twin lane.
        float* red = red_lds;                                 // [WM][BN][2]
        float* gacc = red + WM * BN * 2;                      // [64][2]
        const int b_lo = m0 / p.Mper;
        const int b_hi = min(p.M - 1, m0 + BMt - 1) / p.Mper;
        for (int sb = b_lo; sb <= b_hi; ++sb) {
            const int r_lo = sb * p.Mper, r_hi = min(r_lo + p.Mper, p.M);
            __syncthreads();
            if (tid < 128) gacc[tid] = 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float gs[1][16], gq[1][16];                    // one 32-column slab at a time (register budget of the 128-accumulator tiles)
                {
                    const int cbase = n0 + wn * 32 * TN + j * 32 + 4 * fh;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int col = cbase + 8 * (r >> 2) + (r & 3);
                        const float bias = (p.bias && col < p.N) ? p.bias[col] : 0.f;
                        float s_ = 0.f, q_ = 0.f;
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            const int row = m0 + wm * 32 * TM + i * 32 + fr;
                            const float v = acc[i][j][r] + bias;
                            const bool in = row >= r_lo && row < r_hi;
                            s_ += in ? v : 0.f;
                            q_ += in ? v * v : 0.f;
                        }
                        gs[0][r] = s_;
                        gq[0][r] = q_;
                    }
                }
                float s8[8], q8[8];
                {
                    const bool up = (lane >> 4) & 1;
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        float ss = up ? gs[0][r] : gs[0][r + 8], sq = up ? gq[0][r] : gq[0][r + 8];
                        float ks = up ? gs[0][r + 8] : gs[0][r], kq = up ? gq[0][r + 8] : gq[0][r];
                        s8[r] = ks + __shfl_xor(ss, 16, 64);
                        q8[r] = kq + __shfl_xor(sq, 16, 64);
                    }
                }
                float s4[4], q4[4];
                {
                    const bool up = (lane >> 3) & 1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float ss = up ? s8[r] : s8[r + 4], sq = up ? q8[r] : q8[r + 4];
                        float ks = up ? s8[r + 4] : s8[r], kq = up ? q8[r + 4] : q8[r];
                        s4[r] = ks + __shfl_xor(ss, 8, 64);
                        q4[r] = kq + __shfl_xor(sq, 8, 64);
                    }
                }
                float s2[2], q2[2];
                {
                    const bool up = (lane >> 2) & 1;
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        float ss = up ? s4[r] : s4[r + 2], sq = up ? q4[r] : q4[r + 2];
                        float ks = up ? s4[r + 2] : s4[r], kq = up ? q4[r + 2] : q4[r];
                        s2[r] = ks + __shfl_xor(ss, 4, 64);
                        q2[r] = kq + __shfl_xor(sq, 4, 64);
                    }
                }
                float s1, q1;
                {
                    const bool up = (lane >> 1) & 1;
                    float ss = up ? s2[0] : s2[1], sq = up ? q2[0] : q2[1];
                    float ks = up ? s2[1] : s2[0], kq = up ? q2[1] : q2[0];
                    s1 = ks + __shfl_xor(ss, 2, 64);
                    q1 = kq + __shfl_xor(sq, 2, 64);
                }
                s1 += __shfl_xor(s1, 1, 64);
                q1 += __shfl_xor(q1, 1, 64);
                if ((lane & 1) == 0) {
                    const int r = (lane >> 1) & 15;
                    const int c = wn * 32 * TN + j * 32 + 4 * fh + (r & 3) + 8 * (r >> 2);
                    red[(wm * BN + c) * 2 + 0] = s1;
                    red[(wm * BN + c) * 2 + 1] = q1;
                }
            }
            __syncthreads();
            const int g_first = n0 / p.cpg;
            if (tid < BN && n0 + tid < p.N) {
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int w_ = 0; w_ < WM; ++w_) { s += red[(w_ * BN + tid) * 2]; q += red[(w_ * BN + tid) * 2 + 1]; }
                int gl = (n0 + tid) / p.cpg - g_first;
                atomicAdd(&gacc[gl * 2], s);
                atomicAdd(&gacc[gl * 2 + 1], q);
            }
            __syncthreads();
            const int nloc = (min(n0 + BN, p.N) - 1) / p.cpg - g_first + 1;
            if (tid < nloc * 2)
                atomicAdd(&p.stats[(size_t)sb * 64 + (g_first + (tid >> 1)) * 2 + (tid & 1)], (double)gacc[tid]);
        }
        }
}

// ---- shared epilogue: lane owns pixel row = m0 + wm*32*TM + i*32 + (lane&31) and, per accumulator quad g,
// channels n0 + wn*32*TN + j*32 + 8g + 4*(lane>>5) + {0,1,2,3} ----
template <int WM, int WN, int TM, int TN, bool STATS>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn,
                                              int lane, int tid, char* smem) {
    constexpr int BN = 32 * TN * WN, BMt = 32 * TM * WM;
    const int fr = lane & 31, fh = lane >> 5;
    const bool vec_ok = (p.N & 3) == 0;
    // the 8/16-wave tiles (register budget 128-256) are only launched with staged stores; the direct path is compiled out
    constexpr bool ONLY_STAGED = WM * WN > 4;
    if (ONLY_STAGED || p.epi) gemm_store_staged<WM, WN, TM, TN, STATS>(p, acc, m0, n0, wm, wn, lane, smem);
#pragma unroll
    for (int i = 0; i < ((ONLY_STAGED || p.epi) ? 0 : TM); ++i) {
        const int row = m0 + wm * 32 * TM + i * 32 + fr;
        const bool rok = row < p.M;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int cbase = n0 + wn * 32 * TN + j * 32 + 4 * fh;
            unsigned pk[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = cbase + 8 * g;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
                if (p.bias) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (col + e < p.N) ? p.bias[col + e] : 0.f;
                }
                if (p.act != ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (col + e >= p.act_col0) ? act_apply(v[e], p.act) : v[e];
                }
                const bool full = vec_ok && col + 3 < p.N;
                if (p.res && rok) {
                    const float* rp = p.res + (size_t)row * p.ldr + col;
                    if (full && (p.ldr & 3) == 0) {
                        f32x4 r4 = *reinterpret_cast<const f32x4*>(rp);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += r4[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (col + e < p.N) v[e] += rp[e];
                    }
                }
                if (p.outF && rok) {
                    const int orow = p.out_hw ? (row / p.out_hw) * p.out_stride + p.out_off + row % p.out_hw : row;
                    float* op = p.outF + (size_t)orow * p.ldf + col;
                    if (full && (p.ldf & 3) == 0) {
                        f32x4 o4 = {v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(op) = o4;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (col + e < p.N) op[e] = v[e];
                    }
                }
                if (p.outB && p.b32 == FMT_F32) {
                    if (rok) {
                        float* op = reinterpret_cast<float*>(p.outB) + (size_t)row * p.ldb + col;
                        if (full && (p.ldb & 3) == 0) {
                            f32x4 o4 = {v[0], v[1], v[2], v[3]};
                            *reinterpret_cast<f32x4*>(op) = o4;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) if (col + e < p.N) op[e] = v[e];
                        }
                    }
                } else if (p.outB && p.b32 == FMT_H2) {
                    if (rok && col + 3 < p.N) act_store4(p.outB, (size_t)row * p.ldb + col, v[0], v[1], v[2], v[3], FMT_H2);   // launch_gemm_h2 guarantees N % 8 == 0
                } else if (p.outB) {
                    bf16x2 lo = {(bf16)v[0], (bf16)v[1]}, hi = {(bf16)v[2], (bf16)v[3]};
                    pk[g][0] = __builtin_bit_cast(unsigned, lo);
                    pk[g][1] = __builtin_bit_cast(unsigned, hi);
                }
            }
            if (p.outB && p.b32 == FMT_BF16) {
                // widen to 16-B stores: after the half-swap lanes <32 hold channels 8g..8g+7 of quad pair (g,g+1),
                // lanes >=32 hold channels 8(g+1)..8(g+1)+7
                const int cb0 = n0 + wn * 32 * TN + j * 32;
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    auto s0 = __builtin_amdgcn_permlane32_swap(pk[g][0], pk[g + 1][0], false, false);
                    auto s1 = __builtin_amdgcn_permlane32_swap(pk[g][1], pk[g + 1][1], false, false);
                    const int col = cb0 + 8 * (g + fh);
                    if (rok) {
                        bf16* op = p.outB + (size_t)row * p.ldb + col;
                        if ((p.ldb & 7) == 0 && col + 7 < p.N && ((reinterpret_cast<uintptr_t>(p.outB) & 15) == 0)) {
                            u32x4 o4 = {s0[0], s1[0], s0[1], s1[1]};
                            *reinterpret_cast<u32x4*>(op) = o4;
                        } else {
                            unsigned w4[4] = {s0[0], s1[0], s0[1], s1[1]};
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                if (col + e < p.N) {
                                    unsigned short h = (unsigned short)(w4[e >> 1] >> (16 * (e & 1)));
                                    reinterpret_cast<unsigned short*>(op)[e] = h;
                                }
                        }
                    }
                }
            }
        }
    }
    if (STATS) gemm_stats<WM, WN, TM, TN>(p, acc, m0, n0, wm, wn, lane, tid, reinterpret_cast<float*>(smem));
}

