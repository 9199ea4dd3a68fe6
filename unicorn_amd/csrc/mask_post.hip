// Row N1 (remainder): mask post-processing of the VOS / MOTS drivers on the device -- integer / byte work, bit-exact against
// oracle/mask_oracle.py:
//   mask_resize_kernel     F.interpolate(mask, scale_factor=1/r, "bilinear", align_corners=False)[:, 0, :H, :W] pasted into a zero
//                          (H, W) map; float output (unicorn_vos.py:146-150) or `> thr` bytes (mot_evaluator.py:804-805)
//   vos_merge_kernel       the same resize fused with the soft aggregation of unicorn_vos.py:99-120: background = prod(1 - p),
//                          argmax over [background, ids] -> (H, W) uint8 id map (never materialises the (H, W, K+1) float64 cube)
//   overlap_free_kernel    mot_evaluator.py:860-865: a pixel stays with the first track that claims it
//   rle_*                  pycocotools rleEncode (column-major runs) + rleToString (mot_evaluator.py:889-892)
// HBM-bound streaming kernels: coalesced along x, one pass over the inputs.
#include "kernels.h"
#include <cstdlib>
#include "mask_interp.h"

// the float arithmetic below must round exactly like the numpy restatement: no fused multiply-add anywhere in this file
#pragma clang fp contract(off)

namespace {
__device__ __forceinline__ float bilerp(const float* m, int Wn, const SrcIdx& sy, const SrcIdx& sx) {      // src_index / bilerp4: mask_interp.h
    return bilerp4(m[(size_t)sy.i0 * Wn + sx.i0], m[(size_t)sy.i0 * Wn + sx.i1], m[(size_t)sy.i1 * Wn + sx.i0], m[(size_t)sy.i1 * Wn + sx.i1], sy, sx);
}

constexpr int MR_RPT = 8;     // output rows per thread
__global__ __launch_bounds__(256) void mask_resize_kernel(const float* __restrict__ m, int N, int Hn, int Wn, float rscale, int ho, int wo,
                                                          int H, int W, float thr, float* __restrict__ outF,
                                                          unsigned char* __restrict__ outU) {
    // RPT output rows per thread at one x: consecutive output rows share their source rows (0.74 source rows per output row at 1080p from
    // 800 x 1280), so the 2 x RPT row reads of the one-row kernel (369 us for 64 masks: bound by the L2 -> L1 traffic of the source, 6x the
    // output bytes) shrink to the ~RPT x 0.74 + 1 distinct ones the L1 still holds; same arithmetic per pixel.
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y0 = blockIdx.y * MR_RPT, n = blockIdx.z;
    if (x >= W) return;
    const float* mn = m + (size_t)n * Hn * Wn;
    SrcIdx sx{};
    if (x < wo) sx = src_index(x, Wn, rscale);
#pragma unroll
    for (int r = 0; r < MR_RPT; ++r) {
        const int y = y0 + r;
        if (y >= H) break;
        float v = 0.f;
        if (y < ho && x < wo) v = bilerp(mn, Wn, src_index(y, Hn, rscale), sx);
        const size_t o = ((size_t)n * H + y) * W + x;
        if (outF) outF[o] = v;
        if (outU) outU[o] = v > thr ? 1 : 0;
    }
}

// CondInst scores -> masks at the ORIGINAL resolution in one pass (the MOTS loop never looks at the network-size maps): the factor-f aligned
// bilinear of the coarse (h, w) sigmoid scores (= condinst_final_kernel, utils/boxes.py:138-146 / dynamic_mask_head.py:159-170) and the 1 / r
// bilinear resize + `> thr` (= mask_resize_kernel, mot_evaluator.py:804-805) are chained through LDS instead of through an
// (n, f h, f w) fp32 tensor in HBM (262 MB written + read back for 64 candidates at 800 x 1280).  A block owns a 64 x 32 output tile of one
// instance: phase 1 evaluates ab_sample() for the window of network-grid samples the tile's bilinear taps touch (each sample once:
// ~1100 samples for 2048 outputs at 1080p), phase 2 interpolates them with bilerp4().  Same inline functions and contraction modes as the
// two kernels it replaces (mask_interp.h): bit-identical outputs.  Windows that do not fit the LDS (strong down-scaling) take the
// sample-per-tap path.
constexpr int CR_TX = 64, CR_TY = 32, CR_LDS = 12288;      // output tile, floats of LDS (48 KiB)
template <bool STAGED>
__global__ __launch_bounds__(256) void condinst_resize_kernel(const float* __restrict__ coarse, int h, int w, int f, float rscale, int ho, int wo,
                                                              int H, int W, float thr, float* __restrict__ outF, unsigned char* __restrict__ outU) {
    extern __shared__ float win[];                    // STAGED: the launcher's bound on a tile's window (<= CR_LDS floats; 4.3 KB at 1080p: 8 blocks per CU)
    const int n = blockIdx.z, x0t = blockIdx.x * CR_TX, y0t = blockIdx.y * CR_TY;
    const float* s = coarse + (size_t)n * h * w;
    const int Hn = f * h, Wn = f * w;
    const int tx = threadIdx.x & (CR_TX - 1), tg = threadIdx.x / CR_TX;          // 64 columns x 4 groups of 8 rows
    const int x = x0t + tx;
    int r0 = 0, c0 = 0, wh = 0, ww = 0;
    const bool any = y0t < ho && x0t < wo;            // block-uniform: the tile holds interpolated pixels at all
    if (any) {
        const int yl = min(y0t + CR_TY, ho) - 1, xl = min(x0t + CR_TX, wo) - 1;
        r0 = src_index(y0t, Hn, rscale).i0;          // src_index is monotonic in d: the taps of the tile lie in [r0, r1] x [c0, c1]
        c0 = src_index(x0t, Wn, rscale).i0;
        wh = src_index(yl, Hn, rscale).i1 - r0 + 1;
        ww = src_index(xl, Wn, rscale).i1 - c0 + 1;
    }
    const bool staged = STAGED && any;              // (the launcher picks STAGED only when every tile's window fits: wh * ww <= CR_LDS)
    if (staged) {      // a thread keeps its window column(s): the x coordinates once, no integer division per sample
        for (int xx = tx; xx < ww; xx += CR_TX) {
            int x0, x1;
            float fx;
            ab_coord(c0 + xx, f, w, x0, x1, fx);
#pragma unroll 4
            for (int yy = tg; yy < wh; yy += 256 / CR_TX) {      // (unrolled: the loads of four samples are in flight together)
                int y0, y1;
                float fy;
                ab_coord(r0 + yy, f, h, y0, y1, fy);
                win[__mul24(yy, ww) + xx] = ab_value(s + __mul24(y0, w), s + __mul24(y1, w), x0, x1, fy, fx);     // (24-bit products: full-rate integer multiplies)
            }
        }
    }
    if (STAGED) {
        // phase 2: a thread owns 4 consecutive x of 2 rows (16 x-groups x 16 row pairs): the four column taps once, one packed 4-byte
        // store per row -- the first version (one x, 8 rows, byte stores) spent ~110 VALU instructions per pixel on source indices and
        // 64-bit output addresses and ran at the speed of the two passes it replaces (440 us for 64 masks at 1080p)
        __syncthreads();
        const int qx = threadIdx.x & 15, qy = threadIdx.x >> 4;
        const int xb = x0t + 4 * qx;
        if (xb >= W) return;
        int co[4], dx[4];
        float wx0[4], wx1[4];
        bool inx[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            inx[e] = xb + e < wo;
            const SrcIdx q = src_index(inx[e] ? xb + e : 0, Wn, rscale);
            co[e] = inx[e] ? q.i0 - c0 : 0; dx[e] = inx[e] ? q.i1 - q.i0 : 0; wx0[e] = q.w0; wx1[e] = q.w1;
        }
        const bool packed = (W & 3) == 0;
        const size_t blk0 = ((size_t)n * H + y0t) * W + x0t;       // block-uniform 64-bit part of the output offset (scalar unit)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int y = y0t + 2 * qy + r;
            if (y >= H) break;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (y < ho) {
                const SrcIdx sy = src_index(y, Hn, rscale);
                const float* p0 = win + __mul24(sy.i0 - r0, ww);
                const float* p1 = win + __mul24(sy.i1 - r0, ww);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (inx[e]) {
                        SrcIdx sxe; sxe.i0 = 0; sxe.i1 = 0; sxe.w0 = wx0[e]; sxe.w1 = wx1[e];
                        v[e] = bilerp4(p0[co[e]], p0[co[e] + dx[e]], p1[co[e]], p1[co[e] + dx[e]], sy, sxe);
                    }
            }
            const size_t o = blk0 + (unsigned)(__mul24(2 * qy + r, W) + 4 * qx);
            if (outF) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (xb + e < W) outF[o + e] = v[e];
            }
            if (outU) {
                if (packed) {
                    *reinterpret_cast<unsigned*>(outU + o) = (v[0] > thr ? 1u : 0u) | (v[1] > thr ? 0x100u : 0u) | (v[2] > thr ? 0x10000u : 0u) | (v[3] > thr ? 0x1000000u : 0u);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (xb + e < W) outU[o + e] = v[e] > thr ? 1 : 0;
                }
            }
        }
        return;
    }
    if (x >= W) return;
    SrcIdx sx{};
    if (x < wo) sx = src_index(x, Wn, rscale);
#pragma unroll
    for (int r = 0; r < CR_TY / 4; ++r) {
        const int y = y0t + tg * (CR_TY / 4) + r;
        if (y >= H) break;
        float v = 0.f;
        if (y < ho && x < wo)
            v = bilerp4(ab_sample(s, h, w, f, src_index(y, Hn, rscale).i0, sx.i0), ab_sample(s, h, w, f, src_index(y, Hn, rscale).i0, sx.i1),
                        ab_sample(s, h, w, f, src_index(y, Hn, rscale).i1, sx.i0), ab_sample(s, h, w, f, src_index(y, Hn, rscale).i1, sx.i1),
                        src_index(y, Hn, rscale), sx);
        const size_t o = ((size_t)n * H + y) * W + x;
        if (outF) outF[o] = v;
        if (outU) outU[o] = v > thr ? 1 : 0;
    }
}

__global__ __launch_bounds__(256) void vos_merge_kernel(const float* __restrict__ probs, const int* __restrict__ prob_ids, int K1, int Hn, int Wn,
                                                        float rscale, int ho, int wo, const unsigned char* __restrict__ init_masks,
                                                        const int* __restrict__ init_ids, int K2, int H, int W,
                                                        unsigned char* __restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const bool inside = y < ho && x < wo;
    SrcIdx sy{}, sx{};
    if (inside) { sy = src_index(y, Hn, rscale); sx = src_index(x, Wn, rscale); }
    // np.argmax over channels [0 = background, id]: strictly greater replaces, on equality the LOWER channel index stays
    float bg = 1.f, best = -1.f;
    int best_id = 0x7fffffff;
    for (int k = 0; k < K1; ++k) {
        const float p = inside ? bilerp(probs + (size_t)k * Hn * Wn, Wn, sy, sx) : 0.f;
        bg = bg * (1.f - p);
        const int id = prob_ids[k];
        if (p > best || (p == best && id < best_id)) { best = p; best_id = id; }
    }
    for (int k = 0; k < K2; ++k) {
        const float p = init_masks[((size_t)k * H + y) * W + x] ? 1.f : 0.f;
        bg = bg * (1.f - p);
        const int id = init_ids[k];
        if (p > best || (p == best && id < best_id)) { best = p; best_id = id; }
    }
    // channels of ids that are absent hold 0 and never beat the background (bg >= 0, index 0 wins ties)
    out[(size_t)y * W + x] = (best > bg) ? (unsigned char)best_id : 0;
}

__global__ __launch_bounds__(256) void overlap_free_kernel(const unsigned char* __restrict__ in, int N, size_t HW, unsigned char* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    unsigned char prev = 0;
    for (int n = 0; n < N; ++n) {
        const unsigned char v = in[(size_t)n * HW + i] ? 1 : 0;
        out[(size_t)n * HW + i] = v & (prev ^ 1);
        prev |= v;
    }
}

// ---- RLE.  Column-major element j = x * h + y of mask n is in[n][y][x].  A run starts at j when t[j] != t[j-1] (t[-1] := 0).
// Pass 1 counts the starts per tile of RLE_TILE elements, pass 2 scans the tile counts (one block per mask), pass 3 writes
// the start positions, pass 4 turns them into counts and per-run string lengths, pass 5 scans those and pass 6 writes chars.
constexpr int RLE_TILE = 2048;     // elements per block (256 threads x 8)
__device__ int block_excl_scan(int v, int* total, int* sh /* [256 / 64 + 1] */) {     // 256-thread exclusive scan
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) sh[wv] = inc;
    __syncthreads();
    int base = 0;
    for (int i = 0; i < wv; ++i) base += sh[i];
    if (total) *total = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return base + inc - v;
}
// pycocotools walks a mask COLUMN-major: element j = (x = j / h, y = j % h).  Read straight from the (h, w) bytes every element is its own
// cache line (169 us per pass for the masks of a 1080p MOTS frame); the encoder therefore first writes a binarised TRANSPOSED copy (w, h) into
// its workspace -- 64 x 128 tiles through LDS, 4-byte accesses on both sides -- and the two passes over the elements read it linearly, 8 bytes
// per thread (28 + 14 + 19 us instead of 169 + 168).
constexpr int TRX = 64, TRY = 128;
__global__ __launch_bounds__(256) void rle_transpose_kernel(const unsigned char* __restrict__ in, int h, int w, size_t pitch, unsigned char* __restrict__ out) {
    __shared__ unsigned char tile[TRY][TRX + 4];                 // +4: rows 68 bytes apart (bank spread for the column reads)
    const int n = blockIdx.z, x0 = blockIdx.x * TRX, y0 = blockIdx.y * TRY;
    const unsigned char* m = in + (size_t)n * h * w;
    unsigned char* o = out + (size_t)n * pitch;
    const bool al = (w & 3) == 0 && (reinterpret_cast<uintptr_t>(in) & 3) == 0;
    for (int u = threadIdx.x; u < TRY * (TRX / 4); u += 256) {   // 4 bytes along x per access
        const int yy = u / (TRX / 4), xx = (u - yy * (TRX / 4)) * 4;
        const int y = y0 + yy, x = x0 + xx;
        unsigned char b[4] = {0, 0, 0, 0};
        if (y < h) {
            if (al && x + 3 < w) { const uchar4 q = *reinterpret_cast<const uchar4*>(m + (size_t)y * w + x); b[0] = q.x; b[1] = q.y; b[2] = q.z; b[3] = q.w; }
            else for (int e = 0; e < 4; ++e) if (x + e < w) b[e] = m[(size_t)y * w + x + e];
        }
        for (int e = 0; e < 4; ++e) tile[yy][xx + e] = b[e] ? 1 : 0;
    }
    __syncthreads();
    const bool alo = (h & 3) == 0 && (pitch & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 3) == 0;
    for (int u = threadIdx.x; u < TRX * (TRY / 4); u += 256) {   // 4 bytes along y per access
        const int xx = u / (TRY / 4), yy = (u - xx * (TRY / 4)) * 4;
        const int x = x0 + xx, y = y0 + yy;
        if (x >= w || y >= h) continue;
        unsigned char* dst = o + (size_t)x * h + y;
        if (alo && y + 3 < h) *reinterpret_cast<uchar4*>(dst) = make_uchar4(tile[yy][xx], tile[yy + 1][xx], tile[yy + 2][xx], tile[yy + 3][xx]);
        else for (int e = 0; e < 4 && y + e < h; ++e) dst[e] = tile[yy + e][xx];
    }
}
// 8 consecutive elements of the transposed copy (positions past the end read as the last element: no start is counted there)
__device__ __forceinline__ void rle_load8(const unsigned char* mt, long a, long j0, unsigned char (&t)[8], unsigned char& prev) {
    prev = j0 > 0 ? mt[j0 - 1] : 0;
    if (j0 + 8 <= a && ((reinterpret_cast<uintptr_t>(mt) + (size_t)j0) & 7) == 0) {
        const unsigned long long q = *reinterpret_cast<const unsigned long long*>(mt + j0);
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = (unsigned char)((q >> (8 * e)) & 0xff);
    } else {
        unsigned char last = prev;
#pragma unroll
        for (int e = 0; e < 8; ++e) { if (j0 + e < a) last = mt[j0 + e]; t[e] = last; }
    }
}
__global__ __launch_bounds__(256) void rle_count_kernel(const unsigned char* __restrict__ mt_all, size_t pitch, long a, int ntiles, int* __restrict__ tile_cnt) {
    __shared__ int sh[5];
    const int n = blockIdx.y, tile = blockIdx.x;
    const unsigned char* mt = mt_all + (size_t)n * pitch;
    const long j0 = (long)tile * RLE_TILE + threadIdx.x * 8;
    int c = 0;
    if (j0 < a) {
        unsigned char t[8], prev;
        rle_load8(mt, a, j0, t, prev);
#pragma unroll
        for (int e = 0; e < 8; ++e) { c += t[e] != prev; prev = t[e]; }
    }
    int total;
    (void)block_excl_scan(c, &total, sh);
    if (threadIdx.x == 0) tile_cnt[(size_t)n * ntiles + tile] = total;
}
// exclusive scan of a per-mask int array of length len: one block per mask, 16 consecutive elements per thread (a serial scan in registers +
// ONE block scan of the thread sums per 4096 elements).  The first version walked 256 elements per block scan: 65 chunks x 3 barriers for the
// 16385 run slots of a mask = 2 x 49 us per MOTS frame, now 5 chunks.
constexpr int SCAN_E = 16;
__global__ __launch_bounds__(256) void rle_scan_kernel(int* __restrict__ v, int len, int stride, int* __restrict__ totals) {
    __shared__ int sh[5];
    __shared__ int carry;
    int* p = v + (size_t)blockIdx.x * stride;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < len; base += 256 * SCAN_E) {
        const int i0 = base + threadIdx.x * SCAN_E;
        int x[SCAN_E], sum = 0;
#pragma unroll
        for (int e = 0; e < SCAN_E; ++e) x[e] = i0 + e < len ? p[i0 + e] : 0;
#pragma unroll
        for (int e = 0; e < SCAN_E; ++e) { const int t = x[e]; x[e] = sum; sum += t; }
        int total;
        const int ex = block_excl_scan(sum, &total, sh);
        const int cbase = carry;
#pragma unroll
        for (int e = 0; e < SCAN_E; ++e)
            if (i0 + e < len) p[i0 + e] = cbase + ex + x[e];
        __syncthreads();
        if (threadIdx.x == 0) carry = cbase + total;
        __syncthreads();
    }
    if (threadIdx.x == 0 && totals) totals[blockIdx.x] = carry;
}
__global__ __launch_bounds__(256) void rle_starts_kernel(const unsigned char* __restrict__ mt_all, size_t pitch, long a, int ntiles, const int* __restrict__ tile_off,
                                                         int max_runs, int* __restrict__ starts) {
    __shared__ int sh[5];
    const int n = blockIdx.y, tile = blockIdx.x;
    const unsigned char* mt = mt_all + (size_t)n * pitch;
    const long j0 = (long)tile * RLE_TILE + threadIdx.x * 8;
    int c = 0;
    unsigned flags = 0;
    if (j0 < a) {
        unsigned char t[8], prev;
        rle_load8(mt, a, j0, t, prev);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (t[e] != prev) { flags |= 1u << e; ++c; }      // (positions past the end repeat the last element: never a start)
            prev = t[e];
        }
    }
    int off = tile_off[(size_t)n * ntiles + tile] + block_excl_scan(c, nullptr, sh);
    for (int e = 0; e < 8; ++e)
        if (flags & (1u << e)) {
            if (off < max_runs) starts[(size_t)n * max_runs + off] = (int)(j0 + e);
            ++off;
        }
}
__device__ __forceinline__ int rle_delta(const int* starts, int nstart, long a, int i, int* cnt_i) {
    // runs: cnt[0] = first start (or a), cnt[i] = start[i] - start[i-1], cnt[nstart] = a - start[nstart-1]
    auto cnt = [&](int q) -> long {
        const long lo = q == 0 ? 0 : starts[q - 1];
        const long hi = q < nstart ? starts[q] : a;
        return hi - lo;
    };
    const long c = cnt(i);
    *cnt_i = (int)c;
    return (int)(c - (i > 2 ? cnt(i - 2) : 0));
}
__device__ __forceinline__ int rle_chars(long x, unsigned char* dst) {      // maskApi.c rleToString inner loop
    int n = 0;
    bool more = true;
    while (more) {
        int c = (int)(x & 0x1f);
        x >>= 5;
        more = (c & 0x10) ? x != -1 : x != 0;
        if (more) c |= 0x20;
        if (dst) dst[n] = (unsigned char)(c + 48);
        ++n;
    }
    return n;
}
__global__ __launch_bounds__(256) void rle_len_kernel(const int* __restrict__ starts, const int* __restrict__ nstarts, long a, int max_runs,
                                                      int* __restrict__ lens, unsigned* __restrict__ counts) {
    const int n = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    const int ns = min(nstarts[n], max_runs);
    if (i > ns) return;
    int c;
    const int x = rle_delta(starts + (size_t)n * max_runs, ns, a, i, &c);
    lens[(size_t)n * (max_runs + 1) + i] = rle_chars(x, nullptr);
    if (counts) counts[(size_t)n * (max_runs + 1) + i] = (unsigned)c;
}
__global__ __launch_bounds__(256) void rle_write_kernel(const int* __restrict__ starts, const int* __restrict__ nstarts, long a, int max_runs,
                                                        const int* __restrict__ offs, int max_chars, unsigned char* __restrict__ out) {
    const int n = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    const int ns = min(nstarts[n], max_runs);
    if (i > ns) return;
    int c;
    const int x = rle_delta(starts + (size_t)n * max_runs, ns, a, i, &c);
    const int o = offs[(size_t)n * (max_runs + 1) + i];
    unsigned char tmp[8];
    const int len = rle_chars(x, tmp);
    if (o + len <= max_chars)
        for (int e = 0; e < len; ++e) out[(size_t)n * max_chars + o + e] = tmp[e];
}
}  // namespace

int launch_mask_resize(const float* masks, int N, int Hn, int Wn, float rscale, int ho, int wo, int H, int W, float thr, float* outF,
                       unsigned char* outU, hipStream_t s) {
    if (N == 0) return 0;
    UNI_REQUIRE(Hn > 0 && Wn > 0 && H > 0 && W > 0 && ho > 0 && wo > 0 && N <= 65535 && H <= 65535, "mask_resize: bad geometry");
    hipLaunchKernelGGL(mask_resize_kernel, dim3(cdiv(W, 256), cdiv(H, MR_RPT), N), dim3(256), 0, s, masks, N, Hn, Wn, rscale, ho, wo, H, W, thr, outF, outU);
    return 0;
}
int launch_condinst_resize(const float* coarse, int N, int h, int w, int f, float rscale, int ho, int wo, int H, int W, float thr, float* outF,
                           unsigned char* outU, hipStream_t s) {
    if (N == 0) return 0;
    UNI_REQUIRE(h > 0 && w > 0 && f >= 1 && H > 0 && W > 0 && ho > 0 && wo > 0 && N <= 65535 && cdiv(H, CR_TY) <= 65535, "condinst_resize: bad geometry");
    // window of a tile: taps of CR_TY (CR_TX) consecutive outputs span at most ceil((CR_TY - 1) * rscale) + 3 source rows (columns)
    const long wh = (long)ceilf((CR_TY - 1) * rscale) + 3, ww = (long)ceilf((CR_TX - 1) * rscale) + 3;
    const dim3 grid(cdiv(W, CR_TX), cdiv(H, CR_TY), N);
    // dynamic LDS = the window bound, not the 48 KiB maximum: a block lives for ~5 dependent L2 round trips (phase 1), so the kernel is bound
    // by the blocks in flight per CU (3 with a static 48 KiB array: 345 us for 64 masks; 8 with 4.3 KiB)
    // (the UNI_CR_DBG ablation switches of round 5 -- no window evaluation / no byte stores -- are gone from the product kernel: a stray
    // environment variable could make the default MOTS path return garbage; their measurements are in profiles/r05_mask_bench.txt)
    if (wh * ww <= CR_LDS) hipLaunchKernelGGL(condinst_resize_kernel<true>, grid, dim3(256), (size_t)(wh * ww) * sizeof(float), s, coarse, h, w, f, rscale, ho, wo, H, W, thr, outF, outU);
    else hipLaunchKernelGGL(condinst_resize_kernel<false>, grid, dim3(256), 0, s, coarse, h, w, f, rscale, ho, wo, H, W, thr, outF, outU);
    return 0;
}
int launch_vos_merge(const float* probs, const int* prob_ids, int K1, int Hn, int Wn, float rscale, int ho, int wo,
                     const unsigned char* init_masks, const int* init_ids, int K2, int H, int W, unsigned char* out, hipStream_t s) {
    UNI_REQUIRE(H > 0 && W > 0 && H <= 65535 && (K1 == 0 || (Hn > 0 && Wn > 0 && ho > 0 && wo > 0)), "vos_merge: bad geometry");
    hipLaunchKernelGGL(vos_merge_kernel, dim3(cdiv(W, 256), H), dim3(256), 0, s, probs, prob_ids, K1, Hn, Wn, rscale, ho, wo, init_masks,
                       init_ids, K2, H, W, out);
    return 0;
}
int launch_overlap_free(const unsigned char* in, int N, int H, int W, unsigned char* out, hipStream_t s) {
    if (N == 0) return 0;
    const size_t hw = (size_t)H * W;
    hipLaunchKernelGGL(overlap_free_kernel, dim3((unsigned)((hw + 255) / 256)), dim3(256), 0, s, in, N, hw, out);
    return 0;
}
static size_t rle_pitch(int H, int W) { return ((size_t)H * W + 15) / 16 * 16; }      // bytes per transposed mask (16-byte aligned: 8-byte reads)
static size_t rle_int_bytes(int N, int H, int W, int max_runs) {                        // the integer arrays, rounded to 16 bytes
    const size_t ntiles = ((size_t)H * W + RLE_TILE - 1) / RLE_TILE;
    return (((size_t)N * (ntiles + (size_t)max_runs + 2 * ((size_t)max_runs + 1) + 2)) * sizeof(int) + 256 + 15) / 16 * 16;
}
size_t rle_workspace_bytes(int N, int H, int W, int max_runs) {
    return rle_int_bytes(N, H, W, max_runs) + (size_t)N * rle_pitch(H, W) + 16;      // + the transposed binarised copy of the masks
}
__global__ void rle_finalize_kernel(const int* __restrict__ nstart, int N, int max_runs, int max_chars, int* __restrict__ out_len,
                                    int* __restrict__ n_runs) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    if (n_runs) n_runs[n] = nstart[n] + 1;
    if (nstart[n] > max_runs || out_len[n] > max_chars) out_len[n] = -1;        // overflow: the caller must retry with larger bounds
}

// masks (N, H, W) {0,1} bytes -> out_chars (N, max_chars) + out_len (N) [string length, or -1 when max_runs / max_chars overflow],
// optional counts (N, max_runs + 1) uint32 + n_runs (N)
int launch_rle_encode(const unsigned char* masks, int N, int H, int W, int max_runs, int max_chars, unsigned char* out_chars,
                      int* out_len, unsigned* counts, int* n_runs, void* ws, size_t ws_bytes, hipStream_t s) {
    if (N == 0) return 0;
    UNI_REQUIRE(H > 0 && W > 0 && max_runs > 0 && max_chars > 0 && N <= 65535, "rle: bad argument");
    UNI_REQUIRE(ws_bytes >= rle_workspace_bytes(N, H, W, max_runs), "rle: workspace too small");
    const long a = (long)H * W;
    const int ntiles = (int)((a + RLE_TILE - 1) / RLE_TILE);
    int* tile_cnt = reinterpret_cast<int*>(ws);                  // [N][ntiles] counts -> exclusive offsets
    int* starts = tile_cnt + (size_t)N * ntiles;                 // [N][max_runs] run start positions (column-major element index)
    int* lens = starts + (size_t)N * max_runs;                   // [N][max_runs + 1] chars per run -> exclusive offsets
    int* nstart = lens + (size_t)N * (max_runs + 1);             // [N]
    const size_t pitch = rle_pitch(H, W);
    unsigned char* mt = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(ws) + rle_int_bytes(N, H, W, max_runs) + 15) & ~uintptr_t(15));   // [N][pitch] transposed copy
    UNI_CHECK_HIP(hipMemsetAsync(lens, 0, (size_t)N * (max_runs + 1) * sizeof(int), s));
    hipLaunchKernelGGL(rle_transpose_kernel, dim3(cdiv(W, TRX), cdiv(H, TRY), N), dim3(256), 0, s, masks, H, W, pitch, mt);
    hipLaunchKernelGGL(rle_count_kernel, dim3(ntiles, N), dim3(256), 0, s, mt, pitch, a, ntiles, tile_cnt);
    hipLaunchKernelGGL(rle_scan_kernel, dim3(N), dim3(256), 0, s, tile_cnt, ntiles, ntiles, nstart);
    hipLaunchKernelGGL(rle_starts_kernel, dim3(ntiles, N), dim3(256), 0, s, mt, pitch, a, ntiles, tile_cnt, max_runs, starts);
    hipLaunchKernelGGL(rle_len_kernel, dim3(cdiv(max_runs + 1, 256), N), dim3(256), 0, s, starts, nstart, a, max_runs, lens, counts);
    hipLaunchKernelGGL(rle_scan_kernel, dim3(N), dim3(256), 0, s, lens, max_runs + 1, max_runs + 1, out_len);
    hipLaunchKernelGGL(rle_write_kernel, dim3(cdiv(max_runs + 1, 256), N), dim3(256), 0, s, starts, nstart, a, max_runs, lens, max_chars, out_chars);
    hipLaunchKernelGGL(rle_finalize_kernel, dim3(cdiv(N, 64)), dim3(64), 0, s, nstart, N, max_runs, max_chars, out_len, n_runs);
    return 0;
}
