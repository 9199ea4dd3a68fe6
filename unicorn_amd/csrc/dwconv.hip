// K1a, LDS-tiled: depthwise 7x7 (+bias) with the LayerNorm SPLIT OFF (convnext.py:43-47).
//
// The fused dwconv7+LN kernels of norm.hip need all C channels of a pixel in one block, so their input window can only
// live in registers / L1 (a halo tile of 8x14 px x 768 ch x 4 B = 344 KB does not fit LDS): they are bound by the vector
// L1 (49 taps, ~1.5 TB/s algorithmic).  Here the LayerNorm is moved to the consumer:
//     LN(x) W1^T + b1 = rstd * (x W1'^T - mean * colsum(W1')) + b1',    W1' = W1 diag(gamma),  b1' = b1 + W1 beta
// so this kernel writes the RAW conv output x (operand format of pwconv1) plus per-pixel (mean, rstd), and the pwconv1
// GEMM epilogue applies the row affine (GemmArgs::rowstat / colsum).  Without the cross-channel dependency the kernel
// walks the channels in 32-wide chunks and stages each chunk's halo tile ONCE in LDS:
//   * block = 13 x 16 output pixels (13 rows: 4 % waste on the 25 / 50 / 100 / 200-row maps), 256 threads = 26 (+2 idle) strips
//     of 8 px along x  x  8 channel quads;
//   * per chunk: (13+6) x (16+6) x 32 ch fp32 halo tile (56 KB with the odd row stride) + the chunk's 49 x 32 weights, both by LDS-DMA
//     (global_load_lds_dwordx4: one 1-KiB piece = 8 halo pixels / taps x 128 B; out-of-image pixels read a zero page = the conv's
//     zero padding); DOUBLE buffered (2 x 62 KiB, one block per CU): chunk k+1 flies while chunk k is multiplied, one barrier per chunk;
//   * a thread slides a 14-pixel register window along x for each of the 7 tap rows: 98 + 49 ds_read_b128 feed 1568 FMAs;
//   * LayerNorm statistics are accumulated per thread over the chunks (shifted by the first chunk's pixel mean, so the
//     variance does not cancel), reduced over the 8 quads of a pixel with three shuffles at the end: deterministic, no atomics.
// HBM traffic = input once (+ halo re-reads, L2 hits) + output once; 2.0x L2->LDS amplification.
#include "kernels.h"
#include <cstdlib>

__device__ u32x4 g_zero_page_dw = {0u, 0u, 0u, 0u};

#define GLDS16D(gptr, lptr)                                                                            \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),            \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

namespace {
constexpr int TH = 13, TW = 16, CH = 32;                // output tile (13 divides 26 / 52 / 104 / 208 ~ the 25 / 50 / 100 / 200-row maps), channels per chunk
constexpr int HH = TH + 6, HWD = TW + 6;                // halo tile: 19 rows x 22 pixels
constexpr int HW_ = HWD + 1;                            // LDS row stride 23 pixels (ODD: vertically adjacent pixels fall into different
                                                        // 128-byte halves of the 256-byte bank row, see the lane mapping below)
constexpr int HP = HH * HW_;                            // 437 pixel slots
constexpr int NP_H = (HP + 7) / 8;                      // 1-KiB DMA pieces of the halo tile (8 pixel slots x 128 B) = 55
constexpr int NP_W = (49 + 7) / 8;                      // ... of the chunk's weights (8 taps x 128 B) = 7
constexpr int NPIECE = NP_H + NP_W;                     // 62
constexpr int LDS_BUF = NPIECE * 1024;                  // 63488 B per buffer, two buffers
}  // namespace

// PXS = pixels per thread strip (along x), NW = waves per block: <8, 4> = 26 strips x 8 quads (fewest LDS reads per output, one wave
// per SIMD), <4, 7> = 52 strips x 8 quads (two waves per SIMD hide each other's LDS latency; LDS and VALU pipes both ~saturated)
template <int PXS, int NW>
__global__ __launch_bounds__(64 * NW) void dwconv7_raw_kernel(DwRawArgs p, int tiles_x, int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware remap: consecutive tiles (neighbours in x, then y) on one XCD share their halos in its L2
    int blk;
    {
        const int nwg = gridDim.x, b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int per_img = tiles_x * tiles_y;
    const int sb = blk / per_img, tix = blk - sb * per_img;
    const int ty0 = (tix / tiles_x) * TH, tx0 = (tix - (tix / tiles_x) * tiles_x) * TW;
    const size_t img0 = (size_t)sb * p.H * p.W;
    const int C = p.C;

    // DMA sources: piece pc = wave + NW i; halo pieces cover pixel slots 8 pc .. 8 pc + 7 (lane -> slot 8 pc + lane / 8, 16-byte piece
    // lane % 8 of its 32 channels); weight pieces cover taps 8 (pc - NP_H) .. + 7 of w[tap][chunk channels]
    constexpr int PPW = (NPIECE + NW - 1) / NW;          // pieces per wave
    long src_off[PPW];
    unsigned valid = 0;                                  // bit i: piece i of this lane has a real source (the weight buffer may lie below x: no sign tricks)
    const char* xbase = reinterpret_cast<const char*>(p.x);
    const long zoff = reinterpret_cast<const char*>(&g_zero_page_dw) - xbase;
    const long woff0 = reinterpret_cast<const char*>(p.w) - xbase;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int pc = wave + NW * i;
        if (pc < NP_H) {
            const int hp = pc * 8 + (lane >> 3);
            const int hy = hp / HW_, hx = hp - hy * HW_;
            const int iy = ty0 + hy - 3, ix = tx0 + hx - 3;
            const bool ok = hp < HP && hx < HWD && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            src_off[i] = ok ? (long)((img0 + (size_t)iy * p.W + ix) * C + (lane & 7) * 4) * 4 : 0;
            valid |= ok ? 1u << i : 0u;
        } else {
            const int tap = (pc - NP_H) * 8 + (lane >> 3);
            const bool ok = pc < NPIECE && tap < 49;
            src_off[i] = ok ? woff0 + (long)((size_t)tap * C + (lane & 7) * 4) * 4 : 0;
            valid |= ok ? 1u << i : 0u;
        }
    }
    auto issue = [&](int ck, int buf) __attribute__((always_inline)) {
        char* dst = smem + buf * LDS_BUF;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int pc = wave + NW * i;
            if (pc < NPIECE) {                           // wave-uniform
                long off = (valid >> i) & 1u ? src_off[i] + (long)ck * (CH * 4) : zoff;
                {   // keep ONE DMA per piece (hipcc otherwise splits the select into exec-masked branches)
                    int lo_ = (int)off, hi_ = (int)(off >> 32);
                    asm volatile("" : "+v"(lo_), "+v"(hi_));
                    off = ((long)hi_ << 32) | (unsigned)lo_;
                }
                GLDS16D(xbase + off, dst + pc * 1024);
            }
        }
    };
    // lane mapping: quad c4 = tid & 7; strip s = tid >> 3 with s bit 0 = row parity, next bit(s) = strip along x, upper bits = row pair.
    // A ds_read_b128 lane group {0-3, 12-15, 20-27} then holds quads 0-3 / 4-7 of strips (s, s+1) and (s+2, s+3): rows r and r+1 ->
    // opposite 128-byte halves of the bank row (odd row stride), the two x strips use the other quad half: 16 distinct 16-byte slots.
    const int c4 = tid & 7, strip = tid >> 3;
    constexpr int XS = TW / PXS, XSB = XS == 2 ? 1 : 2;                      // strips along x and their bit count
    const int r = (strip & 1) + 2 * (strip >> (1 + XSB)), x0 = ((strip >> 1) & (XS - 1)) * PXS;
    const bool active = r < TH;
    const int rr = active ? r : 0;
    const int oy = ty0 + r;
    float sum[PXS], sq[PXS], pivot[PXS];
#pragma unroll
    for (int o = 0; o < PXS; ++o) { sum[o] = 0.f; sq[o] = 0.f; pivot[o] = 0.f; }

    const int nchunk = C / CH;
    f32x4 acc[PXS];                                      // outputs of the chunk just computed (stored one iteration later)
    auto store_chunk = [&](int ck) __attribute__((always_inline)) {
#pragma unroll
        for (int o = 0; o < PXS; ++o) {
            const int ox = tx0 + x0 + o;
            if (active && oy < p.H && ox < p.W)
                act_store4(p.out, (img0 + (size_t)oy * p.W + ox) * C + ck * CH + c4 * 4, acc[o][0], acc[o][1], acc[o][2], acc[o][3], p.fmt);
        }
    };
    issue(0, 0);
#pragma unroll 1
    for (int ck = 0; ck < nchunk; ++ck) {
        const int buf = ck & 1;
        __syncthreads();                                 // vmcnt(0) + barrier: chunk ck landed; everyone is done with buf^1
        if (ck + 1 < nchunk && !(p.dbg & 1)) issue(ck + 1, buf ^ 1);     // next chunk's tile + weights fly while this one is multiplied (dbg 1: ablation)
        if (ck > 0 && !(p.dbg & 4)) store_chunk(ck - 1);                 // the previous chunk's outputs: issued AFTER the barrier so that its vmcnt(0) never
                                                         // waits for freshly issued stores (they drain under this chunk's FMAs)
        const float* halo = reinterpret_cast<const float*>(smem + buf * LDS_BUF);
        const float* wl = halo + NP_H * 256;             // [49][CH]
        const int cbase = ck * CH;
        {
            const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + cbase + c4 * 4);
#pragma unroll
            for (int o = 0; o < PXS; ++o) acc[o] = b;
        }
#pragma unroll 1
        for (int ky = 0; ky < ((p.dbg & 2) ? 1 : 7); ++ky) {     // dbg 2: ablation, one tap row only
            const float* rowp = halo + ((rr + ky) * HW_ + x0) * CH + c4 * 4;
            f32x4 win[PXS + 6];
#pragma unroll
            for (int j = 0; j < PXS + 6; ++j) win[j] = *reinterpret_cast<const f32x4*>(rowp + j * CH);
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(wl + (ky * 7 + kx) * CH + c4 * 4);
#pragma unroll
                for (int o = 0; o < PXS; ++o)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[o][e] = fmaf(w[e], win[o + kx][e], acc[o][e]);
            }
        }
        if (ck == 0) {                                   // pivot = mean of the first chunk's 32 channels of each pixel
#pragma unroll
            for (int o = 0; o < PXS; ++o) {
                float s = acc[o][0] + acc[o][1] + acc[o][2] + acc[o][3];
                s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
                pivot[o] = s * (1.f / CH);
            }
        }
#pragma unroll
        for (int o = 0; o < PXS; ++o)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = acc[o][e] - pivot[o]; sum[o] += d; sq[o] = fmaf(d, d, sq[o]); }
    }
    store_chunk(nchunk - 1);
    const float invC = 1.f / (float)C;
#pragma unroll
    for (int o = 0; o < PXS; ++o) {
        float s = sum[o], q = sq[o];
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64);
        const int ox = tx0 + x0 + o;
        if (c4 == 0 && active && oy < p.H && ox < p.W) {
            const float dm = s * invC;                                     // mean - pivot
            const float var = fmaxf(q * invC - dm * dm, 0.f);
            float2 st = make_float2(pivot[o] + dm, rsqrtf(var + p.eps));
            *reinterpret_cast<float2*>(p.stats + (img0 + (size_t)oy * p.W + ox) * 2) = st;
        }
    }
}

int launch_dwconv7_raw(const DwRawArgs& a, hipStream_t s) {
    UNI_REQUIRE(a.C % CH == 0 && a.C >= CH, "dwconv7_raw: C=%d must be a multiple of %d", a.C, CH);
    static const int dbg = getenv("UNI_DW_DBG") ? atoi(getenv("UNI_DW_DBG")) : 0;
    UNI_REQUIRE(((uintptr_t)a.x & 15) == 0 && ((uintptr_t)a.out & 31) == 0 && a.stats, "dwconv7_raw: alignment / NULL stats");
    const int B = a.B > 0 ? a.B : 1;
    const int tiles_x = cdiv(a.W, TW), tiles_y = cdiv(a.H, TH);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv7_raw_kernel<8, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LDS_BUF);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv7_raw_kernel<4, 7>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LDS_BUF);
        attr_done = true;
    }
    DwRawArgs a2 = a;
    a2.dbg = dbg & 7;
    if (dbg & 8) hipLaunchKernelGGL((dwconv7_raw_kernel<8, 4>), dim3(tiles_x * tiles_y * B), dim3(256), 2 * LDS_BUF, s, a2, tiles_x, tiles_y);
    else hipLaunchKernelGGL((dwconv7_raw_kernel<4, 7>), dim3(tiles_x * tiles_y * B), dim3(448), 2 * LDS_BUF, s, a2, tiles_x, tiles_y);
    return 0;
}
