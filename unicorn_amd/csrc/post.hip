// N1: detection post-processing on the device, one image per call --
// unicorn/utils/boxes.py:33-77 (`postprocess`): cxcywh -> corners IN PLACE (:36-39), class max + `obj*cls >= conf_thre`
// filter (:49-55), torchvision `nms` / `batched_nms` (:58-69; greedy, descending score, suppress IoU > thr; batched = per
// class through the coordinate offset `idx * (max_coordinate + 1)`), rows [x1,y1,x2,y2,obj,cls_conf,cls] of the survivors
// in descending-score order.  Everything stays on the device (the reference syncs on every boolean-mask index and inside
// torchvision's NMS); the caller reads back one int32 (the row count) because the result shape is data dependent.
//
// Integer / order semantics are exact: candidates keep their anchor order, the sort is the stable descending argsort
// (rank by counting, ties by anchor index), IoU is evaluated with the same fp32 operation order as the oracle with
// contraction disabled, so kept indices are bit-identical to the CPU restatement.
#include "kernels.h"
#include <algorithm>

namespace {
constexpr int PT = 256;

struct PostWs {                 // layout of the caller-provided workspace
    int* n_cand;                // [1]
    unsigned* maxc_key;         // [1] order-preserving key of the running max coordinate of the candidates
    float* score;               // [A]  obj * cls_conf for candidates, -inf otherwise
    float* cconf;               // [A]
    int* cls;                   // [A]
    int* order;                 // [A]  anchor index of the r-th best candidate
    int* rank;                  // [A]  the candidates' anchor indices, compacted in ascending order
    float* sbox;                // [A][4] class-offset boxes in sorted order
    unsigned long long* mask;   // [A][ceil(A/64)] suppression bits (allocated for n_cand rows only when called)
};

__device__ __forceinline__ unsigned fkey(float f) {     // monotone float -> unsigned
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(unsigned k) {
    unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

__global__ void post_prep_kernel(float* __restrict__ pred, int A, int ld, int nc, float conf_thre, int corners, float* score,
                                 float* cconf, int* cls, int* n_cand, unsigned* maxc_key) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= A) return;
    float* p = pred + (size_t)a * ld;
    float x1 = p[0], y1 = p[1], x2 = p[2], y2 = p[3];
    if (!corners) {                                      // boxes.py:36-39, in place
        const float cx = x1, cy = y1, w = x2, h = y2;
        x1 = cx - w / 2; y1 = cy - h / 2; x2 = cx + w / 2; y2 = cy + h / 2;
        p[0] = x1; p[1] = y1; p[2] = x2; p[3] = y2;
    }
    float best = p[5];
    int bi = 0;
    for (int c = 1; c < nc; ++c) {                       // torch.max: first maximal index
        const float v = p[5 + c];
        if (v > best) { best = v; bi = c; }
    }
    const float sc = p[4] * best;
    const bool keep = sc >= conf_thre;                   // boxes.py:52
    score[a] = keep ? sc : -INFINITY;
    cconf[a] = best;
    cls[a] = bi;
    if (keep) {
        atomicAdd(n_cand, 1);
        const float m = fmaxf(fmaxf(x1, y1), fmaxf(x2, y2));
        atomicMax(maxc_key, fkey(m));
    }
}

// stable descending order of the candidates: rank(a) = #{b : s_b > s_a or (s_b == s_a and b < a)}, order[rank(a)] = a.
// Only candidates can outrank anybody (everything else scores -inf), so the candidates are first COMPACTED in ascending anchor order (one block
// walks the anchors in chunks: deterministic) and the quadratic comparison runs among n_cand scores instead of A: with 64 candidates of 21000
// anchors the A x A version cost 140-145 us per frame in every tracker loop.  Position order in the compact list = anchor order: the tie-break is
// the position.
__global__ __launch_bounds__(1024) void post_compact_kernel(const float* __restrict__ score, int A, int* __restrict__ cidx) {
    __shared__ int wsum[16];
    __shared__ int base_s;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (int a0 = 0; a0 < A; a0 += 1024) {
        const int a = a0 + threadIdx.x;
        const bool c = a < A && score[a] > -INFINITY;
        const unsigned long long b = __ballot(c);
        const int pos = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wv] = __popcll(b);
        __syncthreads();
        int off = base_s;
        for (int i = 0; i < wv; ++i) off += wsum[i];
        if (c) cidx[off + pos] = a;
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
            for (int i = 0; i < 16; ++i) t += wsum[i];
            base_s += t;
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(PT) void post_rank_kernel(const float* __restrict__ score, const int* __restrict__ cidx, const int* __restrict__ n_cand,
                                                       int* __restrict__ order) {
    __shared__ __attribute__((aligned(16))) float tile[PT];
    const int n = *n_cand;
    if ((int)(blockIdx.x * PT) >= n) return;                     // the grid covers the worst case (every anchor a candidate)
    const int c = blockIdx.x * PT + threadIdx.x;
    const bool cand = c < n;
    const int a = cand ? cidx[c] : 0;
    const float sa = cand ? score[a] : INFINITY;
    int cnt = 0;
    for (int t0 = 0; t0 < n; t0 += PT) {
        __syncthreads();
        tile[threadIdx.x] = (t0 + (int)threadIdx.x < n) ? score[cidx[t0 + threadIdx.x]] : -INFINITY;
        __syncthreads();
        // the whole tile in 16-byte broadcast reads, unrolled: entries past the list hold -inf and count for nobody
#pragma unroll 8
        for (int j4 = 0; j4 < PT / 4; ++j4) {
            const float4 t = reinterpret_cast<const float4*>(tile)[j4];
            const int b = t0 + 4 * j4;
            cnt += (t.x > sa) || (t.x == sa && b < c);
            cnt += (t.y > sa) || (t.y == sa && b + 1 < c);
            cnt += (t.z > sa) || (t.z == sa && b + 2 < c);
            cnt += (t.w > sa) || (t.w == sa && b + 3 < c);
        }
    }
    if (cand) order[cnt] = a;
}

__global__ void post_sortbox_kernel(const float* __restrict__ pred, int ld, const int* __restrict__ order,
                                    const int* __restrict__ cls, const int* n_cand, const unsigned* maxc_key,
                                    int class_agnostic, float* __restrict__ sbox) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= *n_cand) return;
    const int a = order[r];
    const float* p = pred + (size_t)a * ld;
    {
#pragma clang fp contract(off)
        // torchvision batched_nms coordinate trick: off = idx * (max_coordinate + 1), fp32, no FMA
        const float off = class_agnostic ? 0.f : (float)cls[a] * (fkey_inv(*maxc_key) + 1.f);
        sbox[4 * r + 0] = p[0] + off;
        sbox[4 * r + 1] = p[1] + off;
        sbox[4 * r + 2] = p[2] + off;
        sbox[4 * r + 3] = p[3] + off;
    }
}

// bit (i, j) = IoU(box_i, box_j) > thr for j > i; one 64 x 64 tile per block
__global__ void post_iou_kernel(const float* __restrict__ sbox, const int* n_cand, float thr, int words,
                                unsigned long long* __restrict__ mask) {
#pragma clang fp contract(off)                            // same fp32 operation order as the oracle, no FMA
    const int n = *n_cand;
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    if (i0 >= n || j0 >= n || j0 + 63 < i0) return;
    __shared__ float cb[64][4];
    const int t = threadIdx.x;
    if (j0 + t < n) {
        cb[t][0] = sbox[4 * (j0 + t)]; cb[t][1] = sbox[4 * (j0 + t) + 1];
        cb[t][2] = sbox[4 * (j0 + t) + 2]; cb[t][3] = sbox[4 * (j0 + t) + 3];
    }
    __syncthreads();
    const int i = i0 + t;
    if (i >= n) return;
    const float x1 = sbox[4 * i], y1 = sbox[4 * i + 1], x2 = sbox[4 * i + 2], y2 = sbox[4 * i + 3];
    const float ai = (x2 - x1) * (y2 - y1);
    unsigned long long bits = 0ull;
    const int lim = min(64, n - j0);
    for (int j = 0; j < lim; ++j) {
        if (j0 + j <= i) continue;
        const float lx = fmaxf(x1, cb[j][0]), ly = fmaxf(y1, cb[j][1]);
        const float rx = fminf(x2, cb[j][2]), ry = fminf(y2, cb[j][3]);
        const float w = fmaxf(rx - lx, 0.f), h = fmaxf(ry - ly, 0.f);
        const float inter = w * h;
        const float aj = (cb[j][2] - cb[j][0]) * (cb[j][3] - cb[j][1]);
        const float iou = inter / (ai + aj - inter);
        if (iou > thr) bits |= 1ull << j;
    }
    mask[(size_t)i * words + blockIdx.x] = bits;
}

// greedy sweep (one block), 64 candidates per step: wave 0 resolves the chunk's internal order from the 64 diagonal words
// (register loop with lane broadcasts, no barriers), every lane of it writes its own surviving row, then all threads OR the
// rows of the chunk's survivors into the removed set of the later chunks.  2 barriers per 64 candidates instead of one per
// candidate (conf 0.001 keeps ~all 21000 anchors as candidates: 21 ms -> < 1 ms).
__global__ __launch_bounds__(1024) void post_sweep_kernel(const float* __restrict__ pred, int ld, const int* __restrict__ order,
                                                          const float* __restrict__ cconf, const int* __restrict__ cls,
                                                          const int* n_cand, int words, const unsigned long long* __restrict__ mask,
                                                          int max_det, float* __restrict__ det, int* __restrict__ keep_idx,
                                                          int* __restrict__ n_out) {
    extern __shared__ unsigned long long removed[];      // [words]
    __shared__ unsigned long long keep_chunk;
    __shared__ int nk;
    const int n = *n_cand;
    const int nw = (n + 63) / 64;
    const int lane = threadIdx.x & 63;
    for (int w = threadIdx.x; w < nw; w += blockDim.x) removed[w] = 0ull;
    if (threadIdx.x == 0) nk = 0;
    for (int c = 0; c < nw; ++c) {
        __syncthreads();                                  // removed[c] is final, nk is current
        if (threadIdx.x < 64) {
            unsigned long long rw = removed[c];
            const int row = c * 64 + lane;
            const unsigned long long diag = row < n ? mask[(size_t)row * words + c] : 0ull;     // bits j > lane of this chunk
            const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
            unsigned long long km = 0ull;
            const int lim = min(64, n - c * 64);
            for (int i = 0; i < lim; ++i) {               // wave-uniform
                const unsigned long long di = ((unsigned long long)__shfl(dhi, i, 64) << 32) | __shfl(dlo, i, 64);
                if (!((rw >> i) & 1ull)) { km |= 1ull << i; rw |= di; }
            }
            const int base = nk;
            if ((km >> lane) & 1ull) {
                const int k = base + __popcll(km & ((1ull << lane) - 1ull));
                if (k < max_det) {
                    const int a = order[row];
                    const float* p = pred + (size_t)a * ld;
                    float* d = det + (size_t)k * 7;
                    d[0] = p[0]; d[1] = p[1]; d[2] = p[2]; d[3] = p[3]; d[4] = p[4];
                    d[5] = cconf[a]; d[6] = (float)cls[a];
                    keep_idx[k] = a;
                }
            }
            if (lane == 0) { keep_chunk = km; nk = base + __popcll(km); }
        }
        __syncthreads();
        const unsigned long long km = keep_chunk;
        for (int w = c + 1 + threadIdx.x; w < nw; w += blockDim.x) {
            unsigned long long acc = removed[w];
            unsigned long long bits = km;
            while (bits) {
                const int i = __ffsll((long long)bits) - 1;
                bits &= bits - 1;
                acc |= mask[(size_t)(c * 64 + i) * words + w];
            }
            removed[w] = acc;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) *n_out = nk < max_det ? nk : max_det;
}
}  // namespace

size_t postprocess_workspace_bytes(int A) {
    const size_t words = (size_t)(A + 63) / 64;
    return 256 + (size_t)A * (4 + 4 + 4 + 4 + 4 + 16) + (size_t)A * words * 8 + 256;
}

int launch_postprocess(float* pred, int A, int ld, int num_classes, float conf_thre, float nms_thre, int flags,
                       int max_det, float* det_out, int32_t* keep_idx, int32_t* n_out, void* ws, size_t ws_bytes,
                       hipStream_t s) {
    UNI_REQUIRE(A >= 0 && num_classes >= 1 && ld >= 5 + num_classes && max_det >= 0, "postprocess: bad shape A=%d nc=%d ld=%d",
                A, num_classes, ld);
    UNI_REQUIRE(ws_bytes >= postprocess_workspace_bytes(A), "postprocess: workspace too small");
    if (A == 0) {
        UNI_CHECK_HIP(hipMemsetAsync(n_out, 0, sizeof(int32_t), s));
        return 0;
    }
    char* base = reinterpret_cast<char*>(ws);
    PostWs w;
    w.n_cand = reinterpret_cast<int*>(base);
    w.maxc_key = reinterpret_cast<unsigned*>(base + 4);
    char* q = base + 256;
    w.score = reinterpret_cast<float*>(q); q += (size_t)A * 4;
    w.cconf = reinterpret_cast<float*>(q); q += (size_t)A * 4;
    w.cls = reinterpret_cast<int*>(q); q += (size_t)A * 4;
    w.order = reinterpret_cast<int*>(q); q += (size_t)A * 4;
    w.rank = reinterpret_cast<int*>(q); q += (size_t)A * 4;
    w.sbox = reinterpret_cast<float*>(q); q += (size_t)A * 16;
    q = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(q) + 7) & ~uintptr_t(7));
    w.mask = reinterpret_cast<unsigned long long*>(q);
    const int words = (A + 63) / 64;
    UNI_CHECK_HIP(hipMemsetAsync(base, 0, 8, s));       // n_cand = 0, maxc_key = 0 (below every real key)
    const int nb = cdiv(A, PT);
    hipLaunchKernelGGL(post_prep_kernel, dim3(nb), dim3(PT), 0, s, pred, A, ld, num_classes, conf_thre, (flags >> 1) & 1, w.score, w.cconf, w.cls,
                       w.n_cand, w.maxc_key);
    hipLaunchKernelGGL(post_compact_kernel, dim3(1), dim3(1024), 0, s, w.score, A, w.rank);       // w.rank holds the compact candidate list
    hipLaunchKernelGGL(post_rank_kernel, dim3(nb), dim3(PT), 0, s, w.score, w.rank, w.n_cand, w.order);
    hipLaunchKernelGGL(post_sortbox_kernel, dim3(nb), dim3(PT), 0, s, pred, ld, w.order, w.cls, w.n_cand, w.maxc_key,
                       flags & 1, w.sbox);
    // the grid covers the worst case (every anchor a candidate); tiles beyond n_cand return immediately
    hipLaunchKernelGGL(post_iou_kernel, dim3(words, words), dim3(64), 0, s, w.sbox, w.n_cand, nms_thre, words, w.mask);
    const size_t lds = (size_t)words * 8;
    hipLaunchKernelGGL(post_sweep_kernel, dim3(1), dim3(1024), lds, s, pred, ld, w.order, w.cconf, w.cls, w.n_cand, words, w.mask,
                       max_det, det_out, keep_idx, n_out);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Input letterbox on the device (row 0 / N1): PreprocessorX.process (external/lib/test/tracker/unicorn_sot.py:111-123) and
// preproc (unicorn/data/data_augment.py:194-214): uint8 HWC image -> resize by r = min(H/h, W/w) with cv2.resize
// INTER_LINEAR semantics for 8-bit data (OpenCV resize.cpp: 11-bit fixed-point coefficients, int32 horizontal pass,
// `(((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2` vertical pass) -> top-left aligned, pad 114, CHW fp32, optional
// RGB<->BGR swap.  One thread per output pixel; integer arithmetic identical to oracle/letterbox_oracle.py.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct Tap { int s0, s1, a0, a1; };
__device__ __forceinline__ Tap lb_tap(int d, double scale, int n, bool clamp_weights) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);      // fx = (float)((dx+0.5)*scale_x - 0.5)
    int s = (int)floorf(f);
    f -= (float)s;
    Tap t;
    if (clamp_weights) {                                     // x: if (sx < 0) fx = 0, sx = 0; if (sx >= w-1) fx = 0, sx = w-1
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= n - 1) { f = 0.f; s = n - 1; }
        t.s0 = s;
        t.s1 = min(s + 1, n - 1);
    } else {                                                 // y: rows are clamped, weights are not
        t.s0 = min(max(s, 0), n - 1);
        t.s1 = min(max(s + 1, 0), n - 1);
    }
    t.a0 = __float2int_rn((1.f - f) * 2048.f);               // saturate_cast<short>: round half to even
    t.a1 = __float2int_rn(f * 2048.f);
    return t;
}
__global__ void letterbox_kernel(const unsigned char* __restrict__ img, int h, int w, int nh, int nw, int swap_rb, int H, int W,
                                 float* __restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    float v[3] = {114.f, 114.f, 114.f};
    if (x < nw && y < nh) {
        const Tap tx = lb_tap(x, (double)w / nw, w, true), ty = lb_tap(y, (double)h / nh, h, false);
        const unsigned char* r0 = img + ((size_t)ty.s0 * w) * 3;
        const unsigned char* r1 = img + ((size_t)ty.s1 * w) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int cs = swap_rb ? 2 - c : c;
            const int h0 = r0[tx.s0 * 3 + cs] * tx.a0 + r0[tx.s1 * 3 + cs] * tx.a1;
            const int h1 = r1[tx.s0 * 3 + cs] * tx.a0 + r1[tx.s1 * 3 + cs] * tx.a1;
            int o = (((ty.a0 * (h0 >> 4)) >> 16) + ((ty.a1 * (h1 >> 4)) >> 16) + 2) >> 2;
            o = min(max(o, 0), 255);
            v[c] = (float)o;
        }
    }
    const size_t plane = (size_t)H * W;
    out[(size_t)y * W + x] = v[0];
    out[plane + (size_t)y * W + x] = v[1];
    out[2 * plane + (size_t)y * W + x] = v[2];
}
}  // namespace

namespace {
__global__ void nms_rows_kernel(const float* __restrict__ boxes, const float* __restrict__ scores, int n, float* __restrict__ rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    rows[6 * i + 0] = boxes[4 * i + 0]; rows[6 * i + 1] = boxes[4 * i + 1];
    rows[6 * i + 2] = boxes[4 * i + 2]; rows[6 * i + 3] = boxes[4 * i + 3];
    rows[6 * i + 4] = scores[i];
    rows[6 * i + 5] = 1.f;
}
}  // namespace

size_t nms_workspace_bytes(int n) { return postprocess_workspace_bytes(n) + (size_t)n * (6 + 7) * sizeof(float) + 256; }

// torchvision.ops.nms(boxes xyxy, scores, iou_threshold) (unicorn/utils/boxes.py:58-64 call site): kept indices in
// descending-score order.  Runs the post-processing kernels above on pre-cornered rows [x1,y1,x2,y2,score,1].
int launch_nms(const float* boxes, const float* scores, int n, float iou_thr, int32_t* keep_idx, int32_t* n_out, void* ws,
               size_t ws_bytes, hipStream_t s) {
    UNI_REQUIRE(n >= 0, "nms: n=%d", n);
    UNI_REQUIRE(ws_bytes >= nms_workspace_bytes(n), "nms: workspace too small");
    if (n == 0) {
        UNI_CHECK_HIP(hipMemsetAsync(n_out, 0, sizeof(int32_t), s));
        return 0;
    }
    char* base = reinterpret_cast<char*>(ws);
    const size_t off = (postprocess_workspace_bytes(n) + 255) & ~(size_t)255;
    float* rows = reinterpret_cast<float*>(base + off);
    float* det = rows + (size_t)n * 6;
    hipLaunchKernelGGL(nms_rows_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, boxes, scores, n, rows);
    return launch_postprocess(rows, n, 6, 1, -INFINITY, iou_thr, /*class agnostic | corners*/ 3, n, det, keep_idx, n_out, base,
                              postprocess_workspace_bytes(n), s);
}

int launch_letterbox(const unsigned char* img, int h, int w, int swap_rb, int H, int W, float* out, double* r_out, hipStream_t s) {
    UNI_REQUIRE(h > 0 && w > 0 && H > 0 && W > 0, "letterbox: empty image %dx%d -> %dx%d", h, w, H, W);
    const double r = std::min((double)H / h, (double)W / w);          // unicorn_sot.py:116
    const int nh = (int)(h * r), nw = (int)(w * r);                    // int(height * r), int(width * r)
    UNI_REQUIRE(nh >= 1 && nw >= 1 && nh <= H && nw <= W, "letterbox: degenerate resize %dx%d", nh, nw);
    if (r_out) *r_out = r;
    hipLaunchKernelGGL(letterbox_kernel, dim3(cdiv(W, 256), H), dim3(256), 0, s, img, h, w, nh, nw, swap_rb, H, W, out);
    return 0;
}
