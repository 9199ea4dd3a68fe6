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
        auto dot = [&](const float* x, const float* y) { float s = 0.f; for (int d = 0; d < dim; ++d) s += x[d] * y[d]; return s; };
        if (c.match_metric == 2) {                                      // cosine (:174-177): F.normalize eps = 1e-12
            std::vector<float> nd(m), nm(M);
            for (int i = 0; i < m; ++i) nd[i] = std::max(std::sqrt(dot(emb[i], emb[i])), 1e-12f);
            for (int j = 0; j < M; ++j) nm[j] = std::max(std::sqrt(dot(me[j], me[j])), 1e-12f);
            for (int i = 0; i < m; ++i)
                for (int j = 0; j < M; ++j) {
                    float s = 0.f;
                    for (int d = 0; d < dim; ++d) s += (emb[i][d] / nd[i]) * (me[j][d] / nm[j]);
                    sc[(size_t)i * M + j] = s;
                }
        } else {
            std::vector<float> f((size_t)m * M);
            for (int i = 0; i < m; ++i)
                for (int j = 0; j < M; ++j) f[(size_t)i * M + j] = dot(emb[i], me[j]);
            // softmax over dim 1 (d2t) and, for bisoftmax, dim 0 (t2d): exp(x - max) / sum  (:166-173)
            for (int i = 0; i < m; ++i) {
                float mx = -INFINITY, s = 0.f;
                for (int j = 0; j < M; ++j) mx = std::max(mx, f[(size_t)i * M + j]);
                for (int j = 0; j < M; ++j) { float e = std::exp(f[(size_t)i * M + j] - mx); sc[(size_t)i * M + j] = e; s += e; }
                for (int j = 0; j < M; ++j) sc[(size_t)i * M + j] /= s;
            }
            if (c.match_metric == 0) {
                std::vector<float> col(m);
                for (int j = 0; j < M; ++j) {
                    float mx = -INFINITY, s = 0.f;
                    for (int i = 0; i < m; ++i) mx = std::max(mx, f[(size_t)i * M + j]);
                    for (int i = 0; i < m; ++i) { col[i] = std::exp(f[(size_t)i * M + j] - mx); s += col[i]; }
                    for (int i = 0; i < m; ++i) sc[(size_t)i * M + j] = (sc[(size_t)i * M + j] + col[i] / s) / 2;
                }
            }
        }
        if (c.with_cats)                                                // :182-184
            for (int i = 0; i < m; ++i)
                for (int j = 0; j < M; ++j)
                    if (lab[i] != mlab[j]) sc[(size_t)i * M + j] *= 0.f;
        for (int i = 0; i < m; ++i) {                                   // greedy assignment (:186-194)
            int best = 0;
            float conf = sc[(size_t)i * M];
            for (int j = 1; j < M; ++j)
                if (sc[(size_t)i * M + j] > conf) { conf = sc[(size_t)i * M + j]; best = j; }   // torch.max: first maximum
            const int64_t id = mid[best];
            if (conf > c.match_score_thr && id > -1) {
                if (b[i].v[4] > c.obj_score_thr) {
                    ids[i] = id;
                    for (int r = 0; r < m; ++r)
                        if (r != i) sc[(size_t)r * M + best] = 0.f;
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
