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

/* ---- ByteTrack (unicorn/tracker/byte_tracker.py:142-337 BYTETracker + STrack, matching.py:39-51,75-91,173-181,
 * kalman_filter.py:40-225, basetrack.py).  The drivers build it as BYTETracker(args, frame_rate=30) and call
 * update(output_results, img_info, img_size) once per frame (unicorn/evaluators/mot_evaluator.py:149,186,208-212). */
typedef struct uni_byte uni_byte;
typedef struct {
    float track_thresh;   /* args.track_thresh */
    int32_t track_buffer; /* args.track_buffer */
    float match_thresh;   /* args.match_thresh */
    int32_t mot20;        /* args.mot20 */
    int32_t frame_rate;   /* constructor argument, default 30 */
} uni_byte_cfg;
uni_byte* uni_byte_create(const uni_byte_cfg* cfg);
void uni_byte_destroy(uni_byte* t);
/* BYTETracker.update (:156-293).  dets: float32 (n, ld) rows, ld == 5: [x1,y1,x2,y2,score], ld >= 6: score = c4*c5 (:165-171),
 * in network-input coordinates; divided by min(size_h/img_h, size_w/img_w) like the reference (:172-174).
 * Outputs (capacity cap rows): activated tracks in the reference's list order: tlwh (m,4) float64, score (m) float32,
 * track_id (m), *n_out = m (clamped to cap).  Returns 0 or a negative code. */
int uni_byte_update(uni_byte* t, const float* dets, int n, int ld, double img_h, double img_w, double size_h, double size_w,
                    int cap, double* out_tlwh, float* out_score, int64_t* out_id, int* n_out);
/* BaseTrack._count (a process-wide id counter in the reference, basetrack.py:12,35-38) and BaseTrack.clean_id() */
int64_t uni_byte_id_count(void);
void uni_byte_clean_id(void);
int uni_byte_lost(const uni_byte* t, int64_t* ids_out, int capacity);   /* ids of self.lost_stracks (parity tests) */

#ifdef __cplusplus
}
#endif
#endif
