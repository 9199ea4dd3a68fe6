/* unicorn_assoc.h -- C ABI of the native association step.
 *
 * Replaces, behind the reference's own class API, the per-frame python/torch CPU logic of
 *   unicorn/tracker/quasi_dense_embed_tracker.py:11-212  (QuasiDenseEmbedTracker: __init__, update_memo, memo, match)
 * which the reference drivers call once per frame on .cpu() tensors (unicorn/evaluators/mot_evaluator.py:1041-1045,
 * external/qdtrack test_omni.py:124-131).  Plain host C++ (the reference is host code too): no torch types, caller owns
 * every buffer, the library owns only the opaque tracker state.  Not thread safe per handle.
 *
 * Reference-side binding: see INTEGRATION.md ("association").
 */
#ifndef UNICORN_ASSOC_H
#define UNICORN_ASSOC_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct uni_qd uni_qd;

/* constructor arguments of QuasiDenseEmbedTracker (quasi_dense_embed_tracker.py:11-22), same defaults via uni_qd_default_cfg */
typedef struct {
    float init_score_thr, obj_score_thr, match_score_thr;
    int32_t memo_tracklet_frames, memo_backdrop_frames;
    float memo_momentum, nms_conf_thr, nms_backdrop_iou_thr, nms_class_iou_thr;
    int32_t with_cats;
    int32_t match_metric; /* 0 = bisoftmax, 1 = softmax, 2 = cosine */
} uni_qd_cfg;

void uni_qd_default_cfg(uni_qd_cfg* cfg);
/* NULL on invalid configuration (the reference asserts, :23-25,35); message via uni_qd_last_error */
uni_qd* uni_qd_create(const uni_qd_cfg* cfg);
void uni_qd_destroy(uni_qd* t);
const char* uni_qd_last_error(void);

/* QuasiDenseEmbedTracker.match (:137-212).  bboxes (n,5) [x1,y1,x2,y2,score], labels (n), track_feats (n,dim), row-major.
 * Outputs (capacity n): out_bboxes (m,5), out_labels (m), out_ids (m) in descending-score order after duplicate removal,
 * valids (n) = the boolean index over the score-sorted detections (return_index=True), *n_out = m.
 * ids: >= 0 track id, -1 unmatched below init_score_thr (backdrop), -2 matched-but-low-score duplicate (:189-190).
 * Returns 0, or a negative code (dim mismatch with the memory, NULL pointers). */
int uni_qd_match(uni_qd* t, const float* bboxes, const int64_t* labels, const float* track_feats, int n, int dim, int frame_id,
                 float* out_bboxes, int64_t* out_labels, int64_t* out_ids, uint8_t* valids, int* n_out);

/* introspection used by the parity tests (self.num_tracklets, len(self.tracklets), self.empty, tracklet ids in memo order) */
int64_t uni_qd_num_tracklets(const uni_qd* t);
int uni_qd_alive(const uni_qd* t, int64_t* ids_out, int capacity);

#ifdef __cplusplus
}
#endif
#endif
