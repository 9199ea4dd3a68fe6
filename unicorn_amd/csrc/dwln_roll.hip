// ROLLING-WINDOW depthwise 7x7 (+bias) + LayerNorm(C) for the big ConvNeXt maps (convnext.py:30-33,47-49: `dwconv` -> `norm`,
// the A operand of pwconv1).  fp32 NHWC in, operand-format rows out.
//
// dwconv7_lnb_kernel (norm.hip) computes ROWS = 2 (4) output rows per work item and re-reads the 6 halo rows for every item:
// 7 (4.4) input float4 loads per output float4 through the vector memory path, which is what bounds it (2.3 TB/s algorithmic,
// 0.29 of the HBM peak, 9 % of a frame).  Here a thread (4 channels x 4 pixels of one image column strip) walks DOWN the image:
// every input row is loaded ONCE per strip and feeds the seven output rows it touches, whose partial sums live in seven ROTATING
// accumulator sets (7 x 4 px x 4 ch = 112 registers).  Input row r completes output row r - 3, which is LayerNorm'ed, stored and
// its set re-armed with the bias for the output row seven further down.  Loads per output float4: 10 / 4 (the 3-pixel x halo of a
// 4-pixel strip) x (R + 6) / R (R = rows of the column chunk one work item owns) ~ 2.8-3.1 instead of 7.
//   * row loop unrolled by 7: the accumulator set of (input row, tap row) is a compile-time register name;
//   * input rows through one buffer descriptor per row + the pixel in the scalar offset (zero padding = range check, as in
//     dwconv7_lnb_kernel); the NEXT input row is requested before the 392 packed FMAs of this one;
//   * the 49 x C weights resident in LDS; the 7 taps of a tap row sit in 7 registers that are re-loaded IN PLACE for the next tap
//     row right after their last use (one tap = 32 FMA cycles, the reload has six taps of slack);
//   * LayerNorm two-pass per completed row (4 pixels): in-wave reduce-scatter (7 shuffles), cross-wave through 64 B of LDS per
//     wave; a strip that is one wave needs no block barrier;
//   * 16-byte output stores by swapping halves between lane pairs through DPP (f16x2: even lane = 8 hi halves, odd lane = 8 lo).
// Work item = (sample, chunk of R rows, 4-px strip); persistent blocks of 8 waves, S strips side by side, XCD-contiguous item ranges.
#include "kernels.h"

namespace {
constexpr int RPX = 4, RIN = RPX + 6;

template <int C>
__global__ __launch_bounds__(512) void dwconv7_lnr_kernel(DwLnArgs p, int S, int spr, int R, int nchunks, int nitems) {
    constexpr int CG = C / 4;
    constexpr int wps = (CG + 63) >> 6;                          // waves per strip; lanes past CG idle (C = 192 / 384)
    extern __shared__ float lds[];
    float* wl = lds;                                             // [49][C]
    float* red = lds + 49 * C;                                   // [waves][4]
    const int tid = threadIdx.x;
    for (int u = tid; u < 49 * C / 4; u += blockDim.x) reinterpret_cast<f32x4*>(wl)[u] = reinterpret_cast<const f32x4*>(p.w)[u];
    __syncthreads();
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sl = wv / wps, cg0 = (wv - sl * wps) * 64 + lane;
    const bool lane_ok = cg0 < CG;
    const int cg = lane_ok ? cg0 : 0;                            // idle lanes shadow channel group 0 and store nothing
    int first, step, count;
    {
        const int nblk = (nitems + S - 1) / S;
        const int x = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
        const int q = nblk >> 3, r = nblk & 7;
        const int start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
        const int cnt = q + (x < r ? 1 : 0);
        first = start + slot;
        step = nslots;
        count = slot < cnt ? (cnt - slot + nslots - 1) / nslots : 0;
    }
    constexpr float invC = 1.f / (float)C;
    const int rowbytes = p.W * C * 4;
    // bias / gamma / beta are re-read (L1 / L2 hits) once per finished row through buffer descriptors (scalar base + ONE per-lane offset)
    // instead of living in 12 registers / three 64-bit pointers next to the 220 registers of the accumulators, the two input rows and the
    // tap row; the output stores go through a descriptor per output row as well (pixel in the scalar offset, pixels past the row end and
    // rows of inactive strips fall outside num_records and are dropped: no branches, no 64-bit vector addresses)
    const int voff = cg * 16;
    auto ld_vec = [&](const float* base) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, C * 4, 0x00020000);
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0));
    };
    const bool odd = cg & 1;
    const float* wbase = wl + cg * 4;

    // Sum of 4 values over the lanes of a wave.  In-row part (16 lanes) on the VALU through DPP rotations (rot 8 / 4 / 2 / 1: every lane of
    // the row ends with the row total), then two halving steps across the four rows: afterwards every lane holds the wave total of value
    // ((lane >> 5) & 1) * 2 + ((lane >> 4) & 1).  3 cross-lane LDS-path operations instead of 7 shuffles.
    const int idx32 = (lane ^ 32) << 2;
    auto wave_scatter_sum4 = [&](const float (&v)[4]) __attribute__((always_inline)) {
        float t[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = v[r];
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false));    // row_ror:8
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xf, 0xf, false));    // row_ror:4
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x122, 0xf, 0xf, false));    // row_ror:2
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x121, 0xf, 0xf, false));    // row_ror:1
            t[r] = x;
        }
        float a2[2];
        {
            const bool up = (lane >> 5) & 1;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float send = up ? t[r] : t[r + 2];
                const float got = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(idx32, __builtin_bit_cast(int, send)));
                a2[r] = (up ? t[r + 2] : t[r]) + got;
            }
        }
        const bool up = (lane >> 4) & 1;
        const float send = up ? a2[0] : a2[1];
        const float got = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, send), (0x10 << 10) | 0x1f));   // lane ^ 16
        return (up ? a2[1] : a2[0]) + got;
    };
    auto strip_sync = [&]() __attribute__((always_inline)) {
        if (wps == 1) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
        else __syncthreads();
    };
    auto reduce4 = [&](const float (&part)[4], float (&tot)[4]) __attribute__((always_inline)) {
        const float t = wave_scatter_sum4(part);
        int l2 = lane;
        asm volatile("" : "+v"(l2));            // recompute the scratch index here: hoisted out of the row loop it is one live register too many (spill)
        if ((l2 & 15) == 0) red[wv * 4 + (((l2 >> 5) & 1) * 2 + ((l2 >> 4) & 1))] = t;
        strip_sync();
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float t2 = 0.f;
            for (int w_ = 0; w_ < wps; ++w_) t2 += red[(sl * wps + w_) * 4 + o];      // fixed order: deterministic
            tot[o] = t2;
        }
        strip_sync();
    };
    auto xor1 = [](unsigned v) __attribute__((always_inline)) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true); };
    const int eb = p.b32 == FMT_BF16 ? 2 : 4;                    // bytes per output element
    // per-lane byte offset of a lane's 16-byte store inside its pixel: f16x2 -- the even lane of a pair writes the 8 hi halves of the
    // pair's 8-channel group, the odd lane the 8 lo halves; bf16 -- the even lane writes the group of pixel o, the odd lane that of o + 1;
    // fp32 -- every lane its own 4 channels.  Idle lanes get an offset outside every descriptor.
    const int soff_lane = !lane_ok ? 0x7ffffff0 : p.b32 == FMT_H2 ? (cg & ~1) * 16 + (odd ? 16 : 0) : p.b32 == FMT_BF16 ? (cg & ~1) * 8 + (odd ? C * 2 : 0) : cg * 16;

#pragma unroll 1
    for (int it = 0; it < count; ++it) {
        const int item = (first + it * step) * S + sl;           // wave-uniform
        const bool active = sl < S && item < nitems;
        const int ia = active ? item : 0;
        const int sx = ia % spr, t_ = ia / spr;
        const int rc = t_ % nchunks, sb = t_ / nchunks;
        const int x0 = sx * RPX, y0 = rc * R;
        const int yend = min(y0 + R, p.H);                       // output rows [y0, yend) are this item's
        const size_t img0 = (size_t)sb * p.H * p.W;

        auto load_row = [&](f32x4 (&dst)[RIN], int iy) __attribute__((always_inline)) {
            const bool rok = active && iy >= 0 && iy < p.H && iy < yend + 3;      // rows below the chunk's last halo row are never used
            float* rowp = const_cast<float*>(p.x) + (img0 + (size_t)(rok ? iy : 0) * p.W) * C;
#pragma unroll
            for (int j = 0; j < RIN; ++j) {
                const int ix = x0 + j - 3;
                const bool ok = rok && ix >= 0 && ix < p.W;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(rowp, 0, ok ? rowbytes : 0, 0x00020000);
                dst[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, cg * 16, ok ? ix * C * 4 : 0, 0));
            }
        };
        // LayerNorm + store of one completed output row (4 pixels), then the accumulator set is re-armed with the bias
        auto finish = [&](f32x4 (&a)[RPX], int yq) __attribute__((always_inline)) {
            float part[4], tot[4], mean[4];
            const f32x4 gam = ld_vec(p.gamma), bet = ld_vec(p.beta);
#pragma unroll
            for (int o = 0; o < 4; ++o) part[o] = lane_ok ? a[o][0] + a[o][1] + a[o][2] + a[o][3] : 0.f;
            reduce4(part, tot);
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                mean[o] = tot[o] * invC;
                const float d0 = a[o][0] - mean[o], d1 = a[o][1] - mean[o], d2 = a[o][2] - mean[o], d3 = a[o][3] - mean[o];
                part[o] = lane_ok ? d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3 : 0.f;
            }
            reduce4(part, tot);
            const bool st = active && yq < yend;
            // descriptor of the output row from pixel x0 on: the valid pixels of the strip (W - x0 of them at most) are in range
            char* orow = reinterpret_cast<char*>(p.out) + (img0 + (size_t)(st ? yq : 0) * p.W + x0) * C * eb;
            const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(orow, 0, st ? (p.W - x0) * C * eb : 0, 0x00020000);
            if (p.b32 == FMT_H2) {
#pragma unroll
                for (int o = 0; o < RPX; ++o) {
                    const float rs = rsqrtf(tot[o] * invC + p.eps);
                    f16x4 h, l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { f16 hh, ll; h2_split((a[o][e] - mean[o]) * rs * gam[e] + bet[e], hh, ll); h[e] = hh; l[e] = ll; }
                    const u32x2 hu = __builtin_bit_cast(u32x2, h), lu = __builtin_bit_cast(u32x2, l);
                    const unsigned s0 = odd ? hu[0] : lu[0], s1 = odd ? hu[1] : lu[1];
                    const unsigned r0 = xor1(s0), r1 = xor1(s1);
                    const u32x4 ov = odd ? u32x4{r0, r1, lu[0], lu[1]} : u32x4{hu[0], hu[1], r0, r1};
                    __builtin_amdgcn_raw_buffer_store_b128(ov, ors, soff_lane, o * C * 4, 0);
                }
            } else if (p.b32 == FMT_BF16) {
#pragma unroll
                for (int o = 0; o < RPX; o += 2) {
                    const float rs0 = rsqrtf(tot[o] * invC + p.eps), rs1 = rsqrtf(tot[o + 1] * invC + p.eps);
                    bf16x4 b0, b1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        b0[e] = (bf16)((a[o][e] - mean[o]) * rs0 * gam[e] + bet[e]);
                        b1[e] = (bf16)((a[o + 1][e] - mean[o + 1]) * rs1 * gam[e] + bet[e]);
                    }
                    const u32x2 u0 = __builtin_bit_cast(u32x2, b0), u1 = __builtin_bit_cast(u32x2, b1);
                    const unsigned s0 = odd ? u0[0] : u1[0], s1 = odd ? u0[1] : u1[1];
                    const unsigned r0 = xor1(s0), r1 = xor1(s1);
                    const u32x4 ov = odd ? u32x4{r0, r1, u1[0], u1[1]} : u32x4{u0[0], u0[1], r0, r1};
                    __builtin_amdgcn_raw_buffer_store_b128(ov, ors, soff_lane, o * C * 2, 0);      // (the odd lane's pixel o + 1 sits in its lane offset)
                }
            } else {
#pragma unroll
                for (int o = 0; o < RPX; ++o) {
                    const float rs = rsqrtf(tot[o] * invC + p.eps);
                    const f32x4 yv = {(a[o][0] - mean[o]) * rs * gam[0] + bet[0], (a[o][1] - mean[o]) * rs * gam[1] + bet[1],
                                      (a[o][2] - mean[o]) * rs * gam[2] + bet[2], (a[o][3] - mean[o]) * rs * gam[3] + bet[3]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, yv), ors, soff_lane, o * C * 4, 0);
                }
            }
            const f32x4 bias4 = ld_vec(p.bias);
#pragma unroll
            for (int o = 0; o < RPX; ++o) a[o] = bias4;
        };

        f32x4 acc[7][RPX];
        {
            const f32x4 bias4 = ld_vec(p.bias);
#pragma unroll
            for (int q = 0; q < 7; ++q)
#pragma unroll
                for (int o = 0; o < RPX; ++o) acc[q][o] = bias4;
        }
        f32x4 cur[RIN], nxt[RIN], w[7];
        load_row(cur, y0 - 3);
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) w[kx] = *reinterpret_cast<const f32x4*>(wbase + kx * C);      // tap row 0
        const int nsteps = R + 6;                                // input rows y0 - 3 .. y0 + R + 2; the same count for every strip of the block
#pragma unroll 1
        for (int r0 = 0; r0 < nsteps; r0 += 7) {
#pragma unroll
            for (int u = 0; u < 7; ++u) {
                const int r = r0 + u;                            // input row y0 - 3 + r feeds output row y0 + r - ky through tap row ky
                if (r < nsteps) {
                    load_row(nxt, y0 - 3 + r + 1);
#pragma unroll
                    for (int ky = 0; ky < 7; ++ky) {
                        const int q = r - ky;                    // chunk-relative output row; its accumulator set is (u - ky) mod 7
                        f32x4 (&a)[RPX] = acc[(u - ky + 7) % 7];
                        const bool use = q >= 0 && q < R;        // block-uniform
                        const float* wn = wbase + (ky == 6 ? 0 : (ky + 1) * 7) * C;     // the tap row after this one (row 0 again for the next input row)
#pragma unroll
                        for (int kx = 0; kx < 7; ++kx) {
                            if (use) {
#pragma unroll
                                for (int o = 0; o < RPX; ++o)
#pragma unroll
                                    for (int e = 0; e < 4; ++e) a[o][e] = fmaf(w[kx][e], cur[o + kx][e], a[o][e]);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            w[kx] = *reinterpret_cast<const f32x4*>(wn + kx * C);       // in-place reload: needed again seven taps from now
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    // the next row becomes the current one BEFORE the finished row is stored: the wait for `nxt` then covers loads
                    // only (gfx9 counts stores on vmcnt too; the stores of finish() drain under the next row's FMAs)
#pragma unroll
                    for (int j = 0; j < RIN; ++j) cur[j] = nxt[j];
                    if (r >= 6) finish(acc[(u + 1) % 7], y0 + r - 6);                    // output row r - 6 has seen its seven input rows
                }
            }
        }
    }
}

template <int C>
int launch_lnr(const DwLnArgs& a, int S, int spr, int R, int nchunks, int nitems, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv7_lnr_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        attr_done = true;
    }
    constexpr int wps = (C / 4 + 63) / 64;
    const size_t ldsb = (size_t)49 * C * 4 + (size_t)8 * 4 * 4;
    hipLaunchKernelGGL((dwconv7_lnr_kernel<C>), dim3(256), dim3(S * wps * 64), ldsb, s, a, S, spr, R, nchunks, nitems);
    return 0;
}
}  // namespace

// Rows per column chunk: the fewest (block rounds) x (rows walked per item) over the candidates; 0 = the map is too small for this kernel
// (fewer than ~1.5 rounds of items even at the smallest chunk: the 2-row kernels fill the chip better there).
int dwconv7_lnr_plan(int B, int H, int W, int C, int* R_out, int* nchunks_out) {
    if (!(C == 192 || C == 256 || C == 384 || C == 512 || C == 768) || (long)W * C * 4 >= (1L << 30)) return 0;
    const int wps = (C / 4 + 63) / 64, S = 8 / wps;
    const int spr = cdiv(W, RPX);
    long best = -1;
    int bestR = 0;
    for (int nch = 1; nch <= H; ++nch) {
        const int R = cdiv(H, nch);
        if (R < 7) break;
        if (cdiv(H, R) != nch) continue;
        const long items = (long)spr * nch * B, blocks = (items + S - 1) / S;
        if (blocks < 384) continue;                              // at least 1.5 rounds of 256 persistent blocks
        const long rounds = (blocks + 255) / 256;
        const long cost = rounds * (R + 6 + 2);                  // + 2: LayerNorm tail / pipeline fill per item
        if (best < 0 || cost < best) { best = cost; bestR = R; }
    }
    if (best < 0) return 0;
    *R_out = bestR;
    *nchunks_out = cdiv(H, bestR);
    return 1;
}

int launch_dwconv7_lnr(const DwLnArgs& a, int R, int nchunks, hipStream_t s) {
    const int wps = (a.C / 4 + 63) / 64, S = 8 / wps;
    const int spr = cdiv(a.W, RPX), nb = a.B > 0 ? a.B : 1;
    const int nitems = spr * nchunks * nb;
    switch (a.C) {
        case 192: return launch_lnr<192>(a, S, spr, R, nchunks, nitems, s);
        case 256: return launch_lnr<256>(a, S, spr, R, nchunks, nitems, s);
        case 384: return launch_lnr<384>(a, S, spr, R, nchunks, nitems, s);
        case 512: return launch_lnr<512>(a, S, spr, R, nchunks, nitems, s);
        case 768: return launch_lnr<768>(a, S, spr, R, nchunks, nitems, s);
        default: uni_set_error("dwconv7_lnr: C=%d unsupported", a.C); return -1;
    }
}
