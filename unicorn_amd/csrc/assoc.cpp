// Native QuasiDense embedding association (row N2).  Host C++: the reference runs this step on CPU tensors once per frame
// (unicorn/tracker/quasi_dense_embed_tracker.py); detections per frame are O(100), the memory O(1000) x 128 floats, so the
// work is a few hundred kFLOP -- what the native version removes is the python/torch dispatch (~150 tiny ops per frame).
// Arithmetic follows the reference's fp32 operation order (box_iou, softmax as exp(x - max) / sum, momentum update) so
// that decisions (ids, valids) reproduce the reference; file:line citations are into that file.
#include "../../include/unicorn_assoc.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <vector>

namespace {
thread_local char g_err[256] = "";
void set_err(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

struct Box { float v[5]; };

struct Tracklet {                  // :68-75
    int64_t id;
    Box bbox;
    std::vector<float> embed;
    int64_t label;
    int last_frame;
    Box velocity;
    int acc_frame;
};
struct Backdrop {                  // :84-89
    std::vector<Box> bboxes;
    std::vector<float> embeds;     // [n][dim]
    std::vector<int64_t> labels;
};

// torchvision.ops.box_iou element: inter / (area_a + area_b - inter)
inline float iou(const Box& a, const Box& b) {
    const float area_a = (a.v[2] - a.v[0]) * (a.v[3] - a.v[1]);
    const float area_b = (b.v[2] - b.v[0]) * (b.v[3] - b.v[1]);
    const float lx = std::max(a.v[0], b.v[0]), ly = std::max(a.v[1], b.v[1]);
    const float rx = std::min(a.v[2], b.v[2]), ry = std::min(a.v[3], b.v[3]);
    const float w = std::max(rx - lx, 0.f), h = std::max(ry - ly, 0.f);
    const float inter = w * h;
    return inter / (area_a + area_b - inter);
}
// Similarity matrix f[i][j] = <x_i, y_j> (x: m rows, y: M rows, both contiguous [.][dim]); the reference calls torch.mm here.  Every
// f[i][j] is ONE fp32 accumulator running over d in ascending order through std::fma (exactly rounded on every machine: vfmadd in the avx clones, the
// correctly rounded libm fmaf in the baseline clone, so the clones agree bit for bit); the loop nest puts j innermost over a transposed copy of y, so
// the compiler vectorises ACROSS memory rows.  Round 6: FOUR detections per pass (each y load feeds four accumulator sets: 8 FMAs per 2 loads + 4
// broadcasts = the FMA ports are the limit) and fused multiply-adds instead of the mul + add pairs of -ffp-contract=off: 285 -> ~110 us per frame at
// 200 detections x 400 memory rows x 128 dims (rounds 1-3: scalar chains, ~3 ms per frame for 100 x 250 x 256).
__attribute__((target_clones("avx512f", "avx2,fma", "default")))
void dot_matrix(const float* x, int m, const float* y, int M, int dim, float* f) {
    constexpr int JB = 32;                                   // memory rows per block: [d][JB] image of 128 B x dim stays in L1 across all i
    const int nb = (M + JB - 1) / JB;
    std::vector<float> yt((size_t)nb * dim * JB, 0.f);
    for (int j = 0; j < M; ++j) {
        float* dst = yt.data() + (size_t)(j / JB) * dim * JB + (j % JB);
        for (int d = 0; d < dim; ++d) dst[(size_t)d * JB] = y[(size_t)j * dim + d];
    }
    for (int b = 0; b < nb; ++b) {
        const float* yb = yt.data() + (size_t)b * dim * JB;
        const int jn = std::min(JB, M - b * JB);
        int i = 0;
        for (; i + 4 <= m; i += 4) {
            const float* x0 = x + (size_t)i * dim;
            const float* x1 = x0 + dim;
            const float* x2 = x1 + dim;
            const float* x3 = x2 + dim;
            float a0[JB], a1[JB], a2[JB], a3[JB];
            for (int k = 0; k < JB; ++k) { a0[k] = 0.f; a1[k] = 0.f; a2[k] = 0.f; a3[k] = 0.f; }
            for (int d = 0; d < dim; ++d) {
                const float v0 = x0[d], v1 = x1[d], v2 = x2[d], v3 = x3[d];
                const float* yr = yb + (size_t)d * JB;
                for (int k = 0; k < JB; ++k) {
                    a0[k] = std::fma(v0, yr[k], a0[k]);
                    a1[k] = std::fma(v1, yr[k], a1[k]);
                    a2[k] = std::fma(v2, yr[k], a2[k]);
                    a3[k] = std::fma(v3, yr[k], a3[k]);
                }
            }
            for (int k = 0; k < jn; ++k) {
                f[(size_t)i * M + b * JB + k] = a0[k];
                f[(size_t)(i + 1) * M + b * JB + k] = a1[k];
                f[(size_t)(i + 2) * M + b * JB + k] = a2[k];
                f[(size_t)(i + 3) * M + b * JB + k] = a3[k];
            }
        }
        for (; i < m; ++i) {
            const float* x0 = x + (size_t)i * dim;
            float a0[JB];
            for (int k = 0; k < JB; ++k) a0[k] = 0.f;
            for (int d = 0; d < dim; ++d) {
                const float v0 = x0[d];
                const float* yr = yb + (size_t)d * JB;
                for (int k = 0; k < JB; ++k) a0[k] = std::fma(v0, yr[k], a0[k]);
            }
            for (int k = 0; k < jn; ++k) f[(size_t)i * M + b * JB + k] = a0[k];
        }
    }
}
// squared norms of the rows of a [rows][dim] matrix (ascending-d chain, like dot_matrix)
// exp(x) for x <= 0 without a libm call, so that the loops over a similarity row / column vectorise (std::exp was 70 k scalar calls = 1.4 of the
// 1.6 ms of a 200-detection frame): n = round(x log2 e) through the 1.5 * 2^23 trick, r = x - n ln 2 in two pieces, the degree-6 polynomial of the
// Cephes expf, 2^n through the exponent bits.  Relative error < 2e-7 (torch's own softmax uses a vectorised exp of the same class, not libm's).
// Non-finite input: NaN (a NaN embedding row, or inf - inf when a row / column maximum is +inf) propagates as with std::exp; +inf and
// anything > 0 is clamped to 0 (-> 1), -inf to -87 (-> ~1.6e-38) -- the integer conversion below never sees a NaN or an out-of-range value
// (that would be undefined behaviour, and the three clones could then disagree).  Selects, not branches: the loops still vectorise.
static inline float exp_neg(float x0) {
    const bool isnan = x0 != x0;
    float x = isnan ? 0.f : x0;
    x = x < -87.f ? -87.f : x;
    x = x > 0.f ? 0.f : x;
    const float t = x * 1.44269504088896341f;
    const float n = (t + 12582912.f) - 12582912.f;
    float r = x - n * 0.693359375f;
    r = r - n * -2.12194440e-4f;
    float p = 1.9875691500e-4f;
    p = p * r + 1.3981999507e-3f;
    p = p * r + 8.3334519073e-3f;
    p = p * r + 4.1665795894e-2f;
    p = p * r + 1.6666665459e-1f;
    p = p * r + 5.0000001201e-1f;
    p = p * (r * r) + r + 1.f;
    int32_t bits = ((int32_t)n + 127) << 23;
    float sc;
    std::memcpy(&sc, &bits, 4);
    return isnan ? x0 : p * sc;
}

// scores = softmax over dim 1 (d2t) and, for bisoftmax, the mean with the softmax over dim 0 (t2d): exp(x - max) / sum
// (quasi_dense_embed_tracker.py:166-173).  f is overwritten.  Multi-versioned like dot_matrix: every lane does the same operations in the same
// order at any vector width (no contraction), so the clones give identical results.
__attribute__((target_clones("avx512f", "avx2", "default")))
void softmax_scores(float* f, float* sc, int m, int M, bool bi) {
    for (int i = 0; i < m; ++i) {
        const float* fi = f + (size_t)i * M;
        float* si = sc + (size_t)i * M;
        float mx = -INFINITY;
        for (int j = 0; j < M; ++j) mx = std::max(mx, fi[j]);
        for (int j = 0; j < M; ++j) si[j] = exp_neg(fi[j] - mx);          // (vectorises: no reduction in this loop)
        // row sum: 16 interleaved chains (chain k takes j = k mod 16, ascending), folded in a fixed order -- the same additions at any vector width,
        // so the clones agree bit for bit (a single scalar chain is latency-bound: 4 cycles per element, ~100 us per frame at 200 x 400)
        float ch[16];
        for (int k = 0; k < 16; ++k) ch[k] = 0.f;
        int j = 0;
        for (; j + 16 <= M; j += 16)
            for (int k = 0; k < 16; ++k) ch[k] += si[j + k];
        for (int k = 0; j < M; ++j, ++k) ch[k] += si[j];
        for (int w = 8; w >= 1; w >>= 1)
            for (int k = 0; k < w; ++k) ch[k] += ch[k + w];
        const float s = ch[0];
        for (int j2 = 0; j2 < M; ++j2) si[j2] /= s;
    }
    if (!bi) return;
    // the column softmax walks the matrix ROW by row (j innermost: contiguous, vectorised); every column still takes its maximum and its
    // sum over i in ascending order
    std::vector<float> cmx(M, -INFINITY), cs(M, 0.f);
    for (int i = 0; i < m; ++i) {
        const float* fi = f + (size_t)i * M;
        for (int j = 0; j < M; ++j) cmx[j] = std::max(cmx[j], fi[j]);
    }
    for (int i = 0; i < m; ++i) {
        float* fi = f + (size_t)i * M;
        for (int j = 0; j < M; ++j) fi[j] = exp_neg(fi[j] - cmx[j]);
        for (int j = 0; j < M; ++j) cs[j] += fi[j];
    }
    for (int i = 0; i < m; ++i) {
        const float* fi = f + (size_t)i * M;
        float* si = sc + (size_t)i * M;
        for (int j = 0; j < M; ++j) si[j] = (si[j] + fi[j] / cs[j]) / 2;
    }
}

void dot_rows(const float* a, int rows, int dim, float* out) {
    for (int i = 0; i < rows; ++i) {
        float s = 0.f;
        for (int d = 0; d < dim; ++d) s += a[(size_t)i * dim + d] * a[(size_t)i * dim + d];
        out[i] = s;
    }
}
}  // namespace

struct uni_qd {
    uni_qd_cfg cfg;
    int64_t num_tracklets = 0;
    int dim = -1;
    std::vector<Tracklet> tracklets;     // dict insertion order (:41); pop keeps the order of the rest
    std::vector<Backdrop> backdrops;     // newest first (:84)
};

extern "C" {

void uni_qd_default_cfg(uni_qd_cfg* c) {
    c->init_score_thr = 0.8f; c->obj_score_thr = 0.5f; c->match_score_thr = 0.5f;
    c->memo_tracklet_frames = 30; c->memo_backdrop_frames = 1;
    c->memo_momentum = 0.8f; c->nms_conf_thr = 0.5f; c->nms_backdrop_iou_thr = 0.3f; c->nms_class_iou_thr = 0.7f;
    c->with_cats = 1; c->match_metric = 0;
}

uni_qd* uni_qd_create(const uni_qd_cfg* cfg) {
    if (!cfg) { set_err("uni_qd_create: NULL cfg"); return nullptr; }
    if (!(cfg->memo_momentum >= 0.f && cfg->memo_momentum <= 1.f) || cfg->memo_tracklet_frames < 0 ||
        cfg->memo_backdrop_frames < 0 || cfg->match_metric < 0 || cfg->match_metric > 2) {   // asserts :23-25,35
        set_err("uni_qd_create: invalid configuration");
        return nullptr;
    }
    uni_qd* t = new uni_qd();
    t->cfg = *cfg;
    return t;
}
void uni_qd_destroy(uni_qd* t) { delete t; }
const char* uni_qd_last_error(void) { return g_err; }
int64_t uni_qd_num_tracklets(const uni_qd* t) { return t ? t->num_tracklets : -1; }
int uni_qd_alive(const uni_qd* t, int64_t* ids_out, int capacity) {
    if (!t) return -1;
    const int n = (int)t->tracklets.size();
    for (int i = 0; i < n && i < capacity; ++i) ids_out[i] = t->tracklets[i].id;
    return n;
}

int uni_qd_match(uni_qd* t, const float* bboxes_in, const int64_t* labels_in, const float* feats_in, int n, int dim, int frame_id,
                 float* out_bboxes, int64_t* out_labels, int64_t* out_ids, uint8_t* valids_out, int* n_out) {
    if (!t || !n_out || n < 0 || (n > 0 && (!bboxes_in || !labels_in || !feats_in || !out_bboxes || !out_labels || !out_ids || !valids_out))) {
        set_err("uni_qd_match: NULL argument");
        return -1;
    }
    if (t->dim >= 0 && n > 0 && dim != t->dim) {
        set_err("uni_qd_match: embedding dim %d differs from the memory's %d", dim, t->dim);
        return -2;
    }
    const uni_qd_cfg& c = t->cfg;
    // ---- sort by score, descending (:139-142)
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return bboxes_in[5 * a + 4] > bboxes_in[5 * b + 4]; });
    std::vector<Box> sb(n);
    for (int i = 0; i < n; ++i) std::memcpy(sb[i].v, bboxes_in + 5 * order[i], sizeof(Box));
    // ---- duplicate removal for potential backdrops and cross classes (:144-152); compares against ALL earlier boxes
    std::vector<uint8_t> valid(n, 1);
    for (int i = 1; i < n; ++i) {
        const float thr = sb[i].v[4] < c.obj_score_thr ? c.nms_backdrop_iou_thr : c.nms_class_iou_thr;
        for (int j = 0; j < i; ++j)
            if (iou(sb[i], sb[j]) > thr) { valid[i] = 0; break; }
    }
    std::vector<Box> b;
    std::vector<int64_t> lab;
    std::vector<const float*> emb;
    for (int i = 0; i < n; ++i) {
        valids_out[i] = valid[i];
        if (!valid[i]) continue;
        b.push_back(sb[i]);
        lab.push_back(labels_in[order[i]]);
        emb.push_back(feats_in + (size_t)dim * order[i]);
    }
    const int m = (int)b.size();
    std::vector<int64_t> ids(m, -1);                                    // :158
    // ---- match against the memory (:161-194)
    if (m > 0 && !t->tracklets.empty()) {
        // memo (:104-135): tracklets in insertion order, then the backdrops (newest first)
        std::vector<const float*> me;
        std::vector<int64_t> mid, mlab;
        for (const Tracklet& tr : t->tracklets) { me.push_back(tr.embed.data()); mid.push_back(tr.id); mlab.push_back(tr.label); }
        for (const Backdrop& bd : t->backdrops)
            for (size_t k = 0; k < bd.labels.size(); ++k) { me.push_back(bd.embeds.data() + k * t->dim); mid.push_back(-1); mlab.push_back(bd.labels[k]); }
        const int M = (int)me.size();
        std::vector<float> sc((size_t)m * M);
        // contiguous copies of both sides (the memory rows live in separate vectors)
        std::vector<float> X((size_t)m * dim), Y((size_t)M * dim);
        for (int i = 0; i < m; ++i) std::memcpy(X.data() + (size_t)i * dim, emb[i], sizeof(float) * dim);
        for (int j = 0; j < M; ++j) std::memcpy(Y.data() + (size_t)j * dim, me[j], sizeof(float) * dim);
        if (c.match_metric == 2) {                                      // cosine (:174-177): F.normalize eps = 1e-12, then mm
            std::vector<float> nn(std::max(m, M));
            auto normalize = [&](std::vector<float>& A, int rows) {
                dot_rows(A.data(), rows, dim, nn.data());
                for (int i = 0; i < rows; ++i) {
                    const float nrm = std::max(std::sqrt(nn[i]), 1e-12f);
                    for (int d = 0; d < dim; ++d) A[(size_t)i * dim + d] /= nrm;
                }
            };
            normalize(X, m);
            normalize(Y, M);
            dot_matrix(X.data(), m, Y.data(), M, dim, sc.data());
        } else {
            std::vector<float> f((size_t)m * M);
            dot_matrix(X.data(), m, Y.data(), M, dim, f.data());
            // softmax over dim 1 (d2t) and, for bisoftmax, dim 0 (t2d): exp(x - max) / sum  (:166-173)
            softmax_scores(f.data(), sc.data(), m, M, c.match_metric == 0);
        }
        // :182-194.  `scores *= cat_same` (:182-184) and the column clearing of the greedy loop (`scores[:i, memo_ind] = 0; scores[i+1:, memo_ind] = 0`,
        // :190-191) are applied WHILE a row is scanned instead of being written into the matrix: row i sees 0 in a column whose category differs
        // (x * 0.f: a NaN score stays NaN like in the reference) or that an earlier row took -- rows above i are never read again, so a strided
        // column store per match (200 x 200 cache lines per frame) buys nothing.  torch.max semantics: the FIRST maximum.
        std::vector<float> taken(M, 1.f), rowv(M);
        for (int i = 0; i < m; ++i) {
            const float* si = sc.data() + (size_t)i * M;
            if (c.with_cats) {
                const int64_t li = lab[i];
                for (int j = 0; j < M; ++j) rowv[j] = (mlab[j] != li ? si[j] * 0.f : si[j]);
            } else {
                for (int j = 0; j < M; ++j) rowv[j] = si[j];
            }
            for (int j = 0; j < M; ++j) rowv[j] = taken[j] == 0.f ? 0.f : rowv[j];
            int best = 0;
            float conf = rowv[0];
            for (int j = 1; j < M; ++j)
                if (rowv[j] > conf) { conf = rowv[j]; best = j; }
            const int64_t id = mid[best];
            if (conf > c.match_score_thr && id > -1) {
                if (b[i].v[4] > c.obj_score_thr) {
                    ids[i] = id;
                    taken[best] = 0.f;
                } else if (conf > c.nms_conf_thr) {
                    ids[i] = -2;
                }
            }
        }
    }
    // ---- new tracklets (:195-201)
    for (int i = 0; i < m; ++i)
        if (ids[i] == -1 && b[i].v[4] > c.init_score_thr) ids[i] = t->num_tracklets++;
    // ---- update_memo (:48-102)
    if (m > 0 && t->dim < 0) t->dim = dim;
    for (int i = 0; i < m; ++i) {
        if (ids[i] < 0) continue;
        auto it = std::find_if(t->tracklets.begin(), t->tracklets.end(), [&](const Tracklet& tr) { return tr.id == ids[i]; });
        if (it != t->tracklets.end()) {
            Tracklet& tr = *it;
            Box vel;
            const float dt = (float)(frame_id - tr.last_frame);
            for (int k = 0; k < 5; ++k) vel.v[k] = (b[i].v[k] - tr.bbox.v[k]) / dt;
            tr.bbox = b[i];
            for (int d = 0; d < dim; ++d) tr.embed[d] = (1 - c.memo_momentum) * tr.embed[d] + c.memo_momentum * emb[i][d];
            tr.last_frame = frame_id;
            tr.label = lab[i];
            for (int k = 0; k < 5; ++k) tr.velocity.v[k] = (tr.velocity.v[k] * tr.acc_frame + vel.v[k]) / (tr.acc_frame + 1);
            tr.acc_frame += 1;
        } else {
            Tracklet tr;
            tr.id = ids[i]; tr.bbox = b[i]; tr.embed.assign(emb[i], emb[i] + dim); tr.label = lab[i];
            tr.last_frame = frame_id; std::memset(tr.velocity.v, 0, sizeof(Box)); tr.acc_frame = 0;
            t->tracklets.push_back(std::move(tr));
        }
    }
    {   // backdrops (:77-89): unmatched detections that do not overlap an earlier detection of this frame
        Backdrop bd;
        for (int i = 0; i < m; ++i) {
            if (ids[i] != -1) continue;
            bool dup = false;
            for (int j = 0; j < i && !dup; ++j) dup = iou(b[i], b[j]) > c.nms_backdrop_iou_thr;
            if (dup) continue;
            bd.bboxes.push_back(b[i]);
            bd.embeds.insert(bd.embeds.end(), emb[i], emb[i] + dim);
            bd.labels.push_back(lab[i]);
        }
        t->backdrops.insert(t->backdrops.begin(), std::move(bd));
    }
    t->tracklets.erase(std::remove_if(t->tracklets.begin(), t->tracklets.end(),
                                      [&](const Tracklet& tr) { return frame_id - tr.last_frame >= c.memo_tracklet_frames; }),
                       t->tracklets.end());                             // :92-97
    if ((int)t->backdrops.size() > c.memo_backdrop_frames) t->backdrops.pop_back();   // :99-100
    // ---- outputs
    for (int i = 0; i < m; ++i) {
        std::memcpy(out_bboxes + 5 * i, b[i].v, sizeof(Box));
        out_labels[i] = lab[i];
        out_ids[i] = ids[i];
    }
    *n_out = m;
    return 0;
}

}  // extern "C"

// =====================================================================================================================
// ByteTrack (byte_tracker.py, matching.py, kalman_filter.py).  float64 state like the reference (numpy default); the
// detections arrive as float32 and the reference's float32 steps (score product, /scale, tlbr->tlwh, thresholds) are
// reproduced in float32.  lap.lapjv(extend_cost, cost_limit) and cython_bbox.bbox_overlaps are third-party code absent
// offline: implemented from their published semantics (exact assignment on the extended matrix; IoU with the +1 pixel
// convention).
// =====================================================================================================================
#include <memory>
#include <unordered_map>
#include <unordered_set>

namespace {
enum { ST_NEW = 0, ST_TRACKED = 1, ST_LOST = 2, ST_REMOVED = 3 };      // basetrack.py:5-9
int64_t g_byte_count = 0;                                               // BaseTrack._count

struct BTrack {                                                         // STrack (:13-140)
    double tlwh0[4];
    double mean[8], cov[64];
    bool has_mean = false, is_activated = false;
    float score = 0.f;
    int tracklet_len = 0, state = ST_NEW, frame_id = 0, start_frame = 0;
    int64_t track_id = 0;
    void tlwh(double* r) const {
        if (!has_mean) { for (int k = 0; k < 4; ++k) r[k] = tlwh0[k]; return; }
        r[0] = mean[0]; r[1] = mean[1]; r[2] = mean[2]; r[3] = mean[3];
        r[2] *= r[3];
        r[0] -= r[2] / 2; r[1] -= r[3] / 2;
    }
    void tlbr(double* r) const { tlwh(r); r[2] += r[0]; r[3] += r[1]; }
};
typedef std::shared_ptr<BTrack> TP;

void to_xyah(const double* tlwh, double* r) {                            // :117-125
    r[0] = tlwh[0] + tlwh[2] / 2; r[1] = tlwh[1] + tlwh[3] / 2; r[2] = tlwh[2] / tlwh[3]; r[3] = tlwh[3];
}

// ---- Kalman filter: F = I + shift (dt = 1), H = [I4 0], weights 1/20 and 1/160 (kalman_filter.py:40-51)
const double WP = 1.0 / 20, WV = 1.0 / 160;
void kf_initiate(const double* m, double* mean, double* cov) {          // :53-79
    for (int k = 0; k < 4; ++k) { mean[k] = m[k]; mean[4 + k] = 0; }
    const double std[8] = {2 * WP * m[3], 2 * WP * m[3], 1e-2, 2 * WP * m[3], 10 * WV * m[3], 10 * WV * m[3], 1e-5, 10 * WV * m[3]};
    for (int i = 0; i < 64; ++i) cov[i] = 0;
    for (int k = 0; k < 8; ++k) cov[9 * k] = std[k] * std[k];
}
void kf_predict(double* mean, double* cov) {                            // multi_predict row (:110-142)
    const double h = mean[3];
    const double std[8] = {WP * h, WP * h, 1e-2, WP * h, WV * h, WV * h, 1e-5, WV * h};
    for (int k = 0; k < 4; ++k) mean[k] += mean[4 + k];
    double fc[64], out[64];                                             // F C, then (F C) F^T
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) fc[8 * i + j] = cov[8 * i + j] + (i < 4 ? cov[8 * (i + 4) + j] : 0.0);
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) out[8 * i + j] = fc[8 * i + j] + (j < 4 ? fc[8 * i + j + 4] : 0.0);
    for (int i = 0; i < 64; ++i) cov[i] = out[i];
    for (int k = 0; k < 8; ++k) cov[9 * k] += std[k] * std[k];
}
void kf_update(double* mean, double* cov, const double* z) {            // :143-168 (project :88-109)
    const double h = mean[3];
    const double std[4] = {WP * h, WP * h, 1e-1, WP * h};
    double S[16], L[16] = {0};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) S[4 * i + j] = cov[8 * i + j] + (i == j ? std[i] * std[i] : 0.0);
    for (int i = 0; i < 4; ++i)                                         // Cholesky S = L L^T
        for (int j = 0; j <= i; ++j) {
            double s = S[4 * i + j];
            for (int k = 0; k < j; ++k) s -= L[4 * i + k] * L[4 * j + k];
            L[4 * i + j] = i == j ? std::sqrt(s) : s / L[4 * j + j];
        }
    double K[32];                                                       // K (8x4): solve S K^T = (P H^T)^T row by row
    for (int r = 0; r < 8; ++r) {
        double y[4], x[4];
        for (int i = 0; i < 4; ++i) {
            double s = cov[8 * r + i];                                  // (P H^T)[r][i] = P[r][i]
            for (int k = 0; k < i; ++k) s -= L[4 * i + k] * y[k];
            y[i] = s / L[4 * i + i];
        }
        for (int i = 3; i >= 0; --i) {
            double s = y[i];
            for (int k = i + 1; k < 4; ++k) s -= L[4 * k + i] * x[k];
            x[i] = s / L[4 * i + i];
        }
        for (int i = 0; i < 4; ++i) K[4 * r + i] = x[i];
    }
    double inn[4];
    for (int i = 0; i < 4; ++i) inn[i] = z[i] - mean[i];
    for (int r = 0; r < 8; ++r) {
        double s = 0;
        for (int i = 0; i < 4; ++i) s += inn[i] * K[4 * r + i];
        mean[r] += s;
    }
    double KS[32];
    for (int r = 0; r < 8; ++r)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int i = 0; i < 4; ++i) s += K[4 * r + i] * S[4 * i + j];
            KS[4 * r + j] = s;
        }
    for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) {
            double s = 0;
            for (int j = 0; j < 4; ++j) s += KS[4 * r + j] * K[4 * c + j];
            cov[8 * r + c] -= s;
        }
}

// cython_bbox.bbox_overlaps element (+1 convention)
double iou_p1(const double* a, const double* b) {
    const double iw = std::min(a[2], b[2]) - std::max(a[0], b[0]) + 1;
    if (iw <= 0) return 0;
    const double ih = std::min(a[3], b[3]) - std::max(a[1], b[1]) + 1;
    if (ih <= 0) return 0;
    const double ua = (a[2] - a[0] + 1) * (a[3] - a[1] + 1) + (b[2] - b[0] + 1) * (b[3] - b[1] + 1) - iw * ih;
    return iw * ih / ua;
}
std::vector<double> iou_distance(const std::vector<TP>& a, const std::vector<TP>& b) {     // matching.py:75-91
    std::vector<double> d(a.size() * b.size());
    std::vector<double> ab(4 * a.size()), bb(4 * b.size());
    for (size_t i = 0; i < a.size(); ++i) a[i]->tlbr(&ab[4 * i]);
    for (size_t j = 0; j < b.size(); ++j) b[j]->tlbr(&bb[4 * j]);
    for (size_t i = 0; i < a.size(); ++i)
        for (size_t j = 0; j < b.size(); ++j) d[i * b.size() + j] = 1 - iou_p1(&ab[4 * i], &bb[4 * j]);
    return d;
}
void fuse_score(std::vector<double>& cost, size_t nr, const std::vector<TP>& dets) {       // matching.py:173-181
    for (size_t i = 0; i < nr; ++i)
        for (size_t j = 0; j < dets.size(); ++j) cost[i * dets.size() + j] = 1 - (1 - cost[i * dets.size() + j]) * (double)dets[j]->score;
}

// exact linear assignment (shortest augmenting paths with potentials, O(n^3)) on the square matrix c (n x n)
void solve_lap(int n, const std::vector<double>& c, std::vector<int>& row_to_col) {
    const double INF = 1e300;
    std::vector<double> u(n + 1, 0), v(n + 1, 0), minv(n + 1);
    std::vector<int> p(n + 1, 0), way(n + 1, 0);
    std::vector<char> used(n + 1);
    for (int i = 1; i <= n; ++i) {
        p[0] = i;
        int j0 = 0;
        std::fill(minv.begin(), minv.end(), INF);
        std::fill(used.begin(), used.end(), 0);
        do {
            used[j0] = 1;
            const int i0 = p[j0];
            double delta = INF;
            int j1 = 0;
            for (int j = 1; j <= n; ++j)
                if (!used[j]) {
                    const double cur = c[(size_t)(i0 - 1) * n + (j - 1)] - u[i0] - v[j];
                    if (cur < minv[j]) { minv[j] = cur; way[j] = j0; }
                    if (minv[j] < delta) { delta = minv[j]; j1 = j; }
                }
            for (int j = 0; j <= n; ++j)
                if (used[j]) { u[p[j]] += delta; v[j] -= delta; } else minv[j] -= delta;
            j0 = j1;
        } while (p[j0] != 0);
        do { const int j1 = way[j0]; p[j0] = p[j1]; j0 = j1; } while (j0);
    }
    row_to_col.assign(n, -1);
    for (int j = 1; j <= n; ++j) row_to_col[p[j] - 1] = j - 1;
}
// matching.py:39-51 with lap.lapjv(cost, extend_cost=True, cost_limit=thresh)
void linear_assignment(const std::vector<double>& cost, int nr, int nc, double thresh, std::vector<std::pair<int, int>>& matches,
                       std::vector<int>& u_rows, std::vector<int>& u_cols) {
    matches.clear(); u_rows.clear(); u_cols.clear();
    if (nr == 0 || nc == 0) {
        for (int i = 0; i < nr; ++i) u_rows.push_back(i);
        for (int j = 0; j < nc; ++j) u_cols.push_back(j);
        return;
    }
    const int n = nr + nc;
    std::vector<double> ext((size_t)n * n, thresh / 2.0);
    for (int i = nr; i < n; ++i)
        for (int j = nc; j < n; ++j) ext[(size_t)i * n + j] = 0;
    for (int i = 0; i < nr; ++i)
        for (int j = 0; j < nc; ++j) ext[(size_t)i * n + j] = cost[(size_t)i * nc + j];
    std::vector<int> r2c;
    solve_lap(n, ext, r2c);
    std::vector<char> col_used(nc, 0);
    for (int i = 0; i < nr; ++i) {
        if (r2c[i] < nc) { matches.emplace_back(i, r2c[i]); col_used[r2c[i]] = 1; } else u_rows.push_back(i);
    }
    for (int j = 0; j < nc; ++j)
        if (!col_used[j]) u_cols.push_back(j);
}

std::vector<TP> joint_stracks(const std::vector<TP>& a, const std::vector<TP>& b) {        // :296-308
    std::unordered_set<int64_t> seen;
    std::vector<TP> res;
    for (const TP& t : a) { seen.insert(t->track_id); res.push_back(t); }
    for (const TP& t : b)
        if (seen.insert(t->track_id).second) res.push_back(t);
    return res;
}
}  // namespace

struct uni_byte {
    uni_byte_cfg cfg;
    double det_thresh;
    int max_time_lost;
    int frame_id = 0;
    std::vector<TP> tracked, lost;
    std::unordered_set<int64_t> removed_ids;     // self.removed_stracks is only ever consulted by track_id (:281)
};

extern "C" {

uni_byte* uni_byte_create(const uni_byte_cfg* cfg) {
    if (!cfg) { set_err("uni_byte_create: NULL cfg"); return nullptr; }
    uni_byte* t = new uni_byte();
    t->cfg = *cfg;
    t->det_thresh = (double)cfg->track_thresh + 0.1;                                        // :152
    t->max_time_lost = (int)((cfg->frame_rate > 0 ? cfg->frame_rate : 30) / 30.0 * cfg->track_buffer);   // :153-154
    return t;
}
void uni_byte_destroy(uni_byte* t) { delete t; }
int64_t uni_byte_id_count(void) { return g_byte_count; }
void uni_byte_clean_id(void) { g_byte_count = 0; }
int uni_byte_lost(const uni_byte* t, int64_t* ids_out, int capacity) {
    if (!t) return -1;
    for (size_t i = 0; i < t->lost.size() && (int)i < capacity; ++i) ids_out[i] = t->lost[i]->track_id;
    return (int)t->lost.size();
}

int uni_byte_update(uni_byte* t, const float* dets_in, int n, int ld, double img_h, double img_w, double size_h, double size_w,
                    int cap, double* out_tlwh, float* out_score, int64_t* out_id, int* n_out) {
    if (!t || !n_out || n < 0 || (n > 0 && (!dets_in || ld < 5)) || (cap > 0 && (!out_tlwh || !out_score || !out_id))) {
        set_err("uni_byte_update: bad argument");
        return -1;
    }
    t->frame_id += 1;
    std::vector<TP> activated, refind, lost_new;
    std::vector<int64_t> removed_new;
    const float scale = (float)std::min(size_h / img_h, size_w / img_w);                   // :172-174 (float32 array / python float)
    const float thr = t->cfg.track_thresh;
    std::vector<TP> dets, dets2;
    for (int i = 0; i < n; ++i) {
        const float* r = dets_in + (size_t)i * ld;
        const float sc = ld == 5 ? r[4] : r[4] * r[5];                                     // :165-171
        const float x1 = r[0] / scale, y1 = r[1] / scale, x2 = r[2] / scale, y2 = r[3] / scale;
        const bool hi = sc > thr, lo = sc > 0.1f && sc < thr;                              // :176-180
        if (!hi && !lo) continue;
        TP d = std::make_shared<BTrack>();
        d->tlwh0[0] = x1; d->tlwh0[1] = y1; d->tlwh0[2] = (float)(x2 - x1); d->tlwh0[3] = (float)(y2 - y1);   // tlbr_to_tlwh in float32
        d->score = sc;
        (hi ? dets : dets2).push_back(d);
    }
    std::vector<TP> unconfirmed, tracked;                                                  // :196-201
    for (const TP& tr : t->tracked) (tr->is_activated ? tracked : unconfirmed).push_back(tr);
    std::vector<TP> pool = joint_stracks(tracked, t->lost);                                // :204
    for (const TP& tr : pool) {                                                            // multi_predict (:33-45)
        if (tr->state != ST_TRACKED) tr->mean[7] = 0;
        kf_predict(tr->mean, tr->cov);
    }
    auto hit = [&](const TP& tr, const TP& det) {                                          // update (:75-91) / re_activate (:61-73)
        double tl[4], z[4];
        det->tlwh(tl);
        to_xyah(tl, z);
        kf_update(tr->mean, tr->cov, z);
        if (tr->state == ST_TRACKED) { tr->tracklet_len += 1; activated.push_back(tr); }
        else { tr->tracklet_len = 0; refind.push_back(tr); }
        tr->frame_id = t->frame_id;
        tr->state = ST_TRACKED;
        tr->is_activated = true;
        tr->score = det->score;
    };
    std::vector<std::pair<int, int>> matches;
    std::vector<int> u_track, u_det, u_track2, u_det2, u_unc;
    {   // first association (:207-224)
        std::vector<double> d = iou_distance(pool, dets);
        if (!t->cfg.mot20) fuse_score(d, pool.size(), dets);
        linear_assignment(d, (int)pool.size(), (int)dets.size(), t->cfg.match_thresh, matches, u_track, u_det);
        for (auto& m : matches) hit(pool[m.first], dets[m.second]);
    }
    std::vector<TP> r_tracked;                                                             // :234
    for (int i : u_track)
        if (pool[i]->state == ST_TRACKED) r_tracked.push_back(pool[i]);
    {   // second association with the low-score detections (:226-252)
        std::vector<double> d = iou_distance(r_tracked, dets2);
        linear_assignment(d, (int)r_tracked.size(), (int)dets2.size(), 0.5, matches, u_track2, u_det2);
        for (auto& m : matches) hit(r_tracked[m.first], dets2[m.second]);
        for (int i : u_track2)
            if (r_tracked[i]->state != ST_LOST) { r_tracked[i]->state = ST_LOST; lost_new.push_back(r_tracked[i]); }
    }
    std::vector<TP> rest;                                                                  // :255
    for (int i : u_det) rest.push_back(dets[i]);
    {   // unconfirmed tracks (:256-266)
        std::vector<double> d = iou_distance(unconfirmed, rest);
        if (!t->cfg.mot20) fuse_score(d, unconfirmed.size(), rest);
        linear_assignment(d, (int)unconfirmed.size(), (int)rest.size(), 0.7, matches, u_unc, u_det);
        for (auto& m : matches) hit(unconfirmed[m.first], rest[m.second]);
        for (int i : u_unc) { unconfirmed[i]->state = ST_REMOVED; removed_new.push_back(unconfirmed[i]->track_id); }
    }
    for (int i : u_det) {                                                                  // new tracks (:268-274), activate (:47-59)
        const TP& tr = rest[i];
        if (tr->score < (float)t->det_thresh) continue;
        tr->track_id = ++g_byte_count;
        double z[4];
        to_xyah(tr->tlwh0, z);
        kf_initiate(z, tr->mean, tr->cov);
        tr->has_mean = true;
        tr->tracklet_len = 0;
        tr->state = ST_TRACKED;
        tr->is_activated = t->frame_id == 1;
        tr->frame_id = tr->start_frame = t->frame_id;
        activated.push_back(tr);
    }
    for (const TP& tr : t->lost)                                                           // :276-279
        if (t->frame_id - tr->frame_id > t->max_time_lost) { tr->state = ST_REMOVED; removed_new.push_back(tr->track_id); }
    // ---- state update (:283-291)
    std::vector<TP> keep;
    for (const TP& tr : t->tracked)
        if (tr->state == ST_TRACKED) keep.push_back(tr);
    t->tracked = joint_stracks(joint_stracks(keep, activated), refind);
    auto sub_by_ids = [](const std::vector<TP>& a, const std::unordered_set<int64_t>& ids) {   // sub_stracks (:311-320): dict by id
        std::vector<TP> res;
        std::unordered_map<int64_t, size_t> pos;
        for (const TP& x : a) {
            auto it = pos.find(x->track_id);
            if (it == pos.end()) { pos[x->track_id] = res.size(); res.push_back(x); } else res[it->second] = x;
        }
        std::vector<TP> out;
        for (const TP& x : res)
            if (!ids.count(x->track_id)) out.push_back(x);
        return out;
    };
    std::unordered_set<int64_t> tracked_ids;
    for (const TP& tr : t->tracked) tracked_ids.insert(tr->track_id);
    t->lost = sub_by_ids(t->lost, tracked_ids);
    for (const TP& tr : lost_new) t->lost.push_back(tr);
    t->lost = sub_by_ids(t->lost, t->removed_ids);
    for (int64_t id : removed_new) t->removed_ids.insert(id);
    {   // remove_duplicate_stracks (:323-337)
        std::vector<double> pd = iou_distance(t->tracked, t->lost);
        std::vector<char> dupa(t->tracked.size(), 0), dupb(t->lost.size(), 0);
        for (size_t p = 0; p < t->tracked.size(); ++p)
            for (size_t q = 0; q < t->lost.size(); ++q)
                if (pd[p * t->lost.size() + q] < 0.15) {
                    const int tp = t->tracked[p]->frame_id - t->tracked[p]->start_frame;
                    const int tq = t->lost[q]->frame_id - t->lost[q]->start_frame;
                    if (tp > tq) dupb[q] = 1; else dupa[p] = 1;
                }
        std::vector<TP> ra, rb;
        for (size_t p = 0; p < t->tracked.size(); ++p) if (!dupa[p]) ra.push_back(t->tracked[p]);
        for (size_t q = 0; q < t->lost.size(); ++q) if (!dupb[q]) rb.push_back(t->lost[q]);
        t->tracked.swap(ra);
        t->lost.swap(rb);
    }
    int m = 0;
    for (const TP& tr : t->tracked) {                                                      // :293
        if (!tr->is_activated) continue;
        if (m < cap) {
            tr->tlwh(out_tlwh + 4 * m);
            out_score[m] = tr->score;
            out_id[m] = tr->track_id;
        }
        ++m;
    }
    *n_out = m < cap ? m : cap;
    return 0;
}

}  // extern "C"
