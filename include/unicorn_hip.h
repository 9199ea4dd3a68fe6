/* libunicorn_hip.so — C-ABI of the MI355X-native Unicorn inference hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch types.  The caller
 * (Python/ctypes in unicorn_amd/, or any other host) owns every input/output buffer (device memory,
 * e.g. torch.Tensor.data_ptr()); the library owns only an opaque uni_ctx holding re-packed weights and
 * its scratch workspace.  Every call is asynchronous on the hipStream_t passed in (no hidden sync), and
 * thread-compatible (one ctx per stream / process).  Return 0 on success, <0 on error
 * (uni_last_error() gives the message).
 *
 * Activation layout: NHWC ("channels_last"), fp32 at the API, i.e. a (1,C,H,W) torch tensor with
 * channels_last strides is passed as-is.  Images enter as the reference's NCHW fp32 0-255 BGR tensor.
 *
 * What each entry point replaces in the reference (paths relative to MasterBin-IIAU/Unicorn):
 *   uni_msda_fwd            MultiScaleDeformableAttention.ms_deform_attn_forward
 *                           (unicorn/models/ops/src/vision.cpp:13-16, ops/src/ms_deform_attn.h:19-38,
 *                            ops/src/cuda/ms_deform_attn_cuda.cu:20-80)
 *   uni_corr_softmax_pv     simi = E_ref^T E_cur; softmax(dim=0); values @ trans
 *                           (external/lib/test/tracker/unicorn_sot.py:95-100, unicorn_vos.py:166-181)
 *   uni_backbone_fpn        Unicorn.forward(mode="backbone") (unicorn/models/unicorn.py:231-258;
 *                            backbone/convnext.py:141-154, backbone/yolo_pafpn_new.py:113-161)
 *   uni_interaction         Unicorn.forward(mode="interaction") (unicorn.py:260-276,
 *                            deformable_transformer.py:58-131, ops/modules/ms_deform_attn.py:78-115)
 *   uni_upsample            Unicorn.forward(mode="upsample") (unicorn.py:41-44,311-313)
 *   uni_head                UnicornHead.forward / UnicornHeadMask.forward, eval branch
 *                           (unicorn_head.py:249-336,430-482; unicorn_head_mask.py:280-372,451-519;
 *                            condinst/mask_branch.py:77-99,158-162)
 *   uni_condinst_masks      DynamicMaskHead.__call__ + aligned_bilinear(d_rate)
 *                           (condinst/dynamic_mask_head.py:172-225; utils/boxes.py:138-146)
 *   uni_condinst_masks_u8   the same fused with the 1/r resize + `> mask_thres` of the MOTS loop (mot_evaluator.py:804-805)
 *   uni_letterbox           PreprocessorX.process / preproc (unicorn_sot.py:111-123, data/data_augment.py:194-214)
 *   uni_postprocess         postprocess + torchvision nms/batched_nms (utils/boxes.py:33-77)
 *   uni_prior_pyramid       F.interpolate(coarse, 1/2 | 1/4, bilinear) (unicorn_sot.py:103-105)
 *   uni_label_map_s8        get_label_map + F.interpolate(1/8) (unicorn_sot.py:52-53,128-139)
 *   uni_sample_embeddings   per-box F.grid_sample of the embedding map (evaluators/mot_evaluator.py:1024-1034)
 *   uni_pos_embed           PositionEmbeddingLearned.forward (+ identity bicubic) (position_encoding.py:25-36,
 *                            unicorn.py:248-250)
 *   low-level ops (uni_gemm_bf16, uni_layernorm, uni_dwconv7_ln, uni_groupnorm_act, uni_stem)
 *                           building blocks exported for the kernel parity tests.
 */
#ifndef UNICORN_HIP_H
#define UNICORN_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct uni_ctx uni_ctx;
typedef void* uni_stream_t; /* hipStream_t */

/* Network shape, mirrors exp/unicorn_track.py:31-113 + exps/default/ *.py */
typedef struct uni_model_cfg {
    int32_t dims[4];     /* ConvNeXt stage widths: tiny 96,192,384,768 / large 192,384,768,1536 */
    int32_t depths[4];   /* 3,3,9,3 / 3,3,27,3 */
    int32_t num_classes; /* MOT classes (8 BDD100K, 1 MOT17) */
    int32_t mask;        /* 1 = UnicornHeadMask (+ mask branch, controllers) */
    int32_t n_layer_att; /* 3 */
    int32_t embed_dim;   /* 128 */
    int32_t up_rate;     /* 8 // d_rate (4) */
    int32_t d_rate;      /* 2 */
    int32_t precision;   /* operand format of every dense contraction (fp32 accumulate): 0 = bf16; 1 = exact fp32
                            (v_mfma_f32_32x32x2_f32); 2 = "f16x2" split f16, fp32-equivalent (3 f16 MFMAs per product) */
} uni_model_cfg;

const char* uni_last_error(void);
int uni_version(void);

/* ---- context / weights ------------------------------------------------------------------------- */
uni_ctx* uni_ctx_create(int device_id, const uni_model_cfg* cfg);
void uni_ctx_destroy(uni_ctx* ctx);
/* Register one reference state-dict tensor (fp32, HOST memory, reference layout e.g. OIHW).  Unknown
 * names are ignored (returns 1), like load_state_dict(strict=False). */
int uni_ctx_load_param(uni_ctx* ctx, const char* name, const float* host_data, const int64_t* shape, int ndim);
/* Deployment artefact for hosts without Python / torch (round 6; the role `tools/export_torchscript.py:51-71` plays for the reference: a file
 * another runtime loads): a FLAT WEIGHTS FILE written by `unicorn_amd.utils.checkpoint.export_flat` (or tools/export_weights.py) --
 * "UNIW1\0\0\0", the uni_model_cfg (15 int32), int32 tensor count, then per tensor: int32 name length, name bytes, int32 ndim, int64 shape[ndim],
 * fp32 data (reference layout, little endian).  uni_weights_file_cfg reads the configuration (precision as exported; the caller may change it before
 * uni_ctx_create), uni_ctx_load_file registers every tensor of the file like uni_ctx_load_param (call uni_ctx_finalize afterwards);
 * *n_loaded = tensors read.  tools/capi_host_demo.cpp runs the SOT step from such a file with nothing but this header and the HIP runtime. */
int uni_weights_file_cfg(const char* path, uni_model_cfg* cfg_out);
int uni_ctx_load_file(uni_ctx* ctx, const char* path, int* n_loaded);
/* Re-pack weights for the device (NHWC / [N][K] bf16, layer-scale folded).  *n_missing = parameters the
 * configured network needs but which were never loaded (left at zero). */
int uni_ctx_finalize(uni_ctx* ctx, int* n_missing);
/* name of the i-th missing parameter after finalize (NULL when out of range) */
const char* uni_ctx_missing_name(uni_ctx* ctx, int i);
/* Pre-size the scratch workspace for an (H,W) input (optional; grows on demand otherwise). */
int uni_ctx_reserve(uni_ctx* ctx, int B, int H, int W);

/* Operand-range check of the "f16x2" mode.  Every fp32 value entering a contraction is stored as hi + lo f16 halves and SATURATES at
 * +-65504 (csrc/common.h h2_split; never inf / NaN).  The synthetic test weights stay far below that, a trained checkpoint cannot be
 * validated offline: uni_ctx_set_check(ctx, 1) (or UNI_CHECK_SAT=1 in the environment at finalize) makes the context scan every
 * operand buffer it produces (one extra pass each; the fused MLP falls back to its two-launch form so the hidden activations exist)
 * and count saturated elements.  uni_ctx_stats synchronises and fills out4 = {saturated operands, operands scanned, buffers scanned,
 * 0} since the last uni_ctx_set_check call. */
int uni_ctx_set_check(uni_ctx* ctx, int on);
int uni_ctx_stats(uni_ctx* ctx, long long* out4);

/* Per-kernel-class timing with HIP events on the launch stream (used by bench.py's roofline leg, off by default).
 * uni_prof_end synchronises the device and fills out16: [5 classes][ms, work, launches] + out16[15] = algorithmic
 * bytes of the GEMM class; classes: 0 GEMM/conv (work = algorithmic FLOPs 2*M*N*K), 1 dwconv7+LN, 2 GroupNorm apply,
 * 3 LayerNorm (work = algorithmic bytes), 4 misc. */
int uni_prof_begin(uni_ctx* ctx);
int uni_prof_end(uni_ctx* ctx, double* out16);

/* ---- stage entry points (one per Unicorn.forward mode) ----------------------------------------------- */
/* Every stage takes a batch B of images (the reference modules are batched too; its inference drivers use B = 1).
 * img: (B,3,H,W) fp32 NCHW.  fpn{0,1,2}: NHWC fp32 (B,H/8,W/8,C1), (B,H/16,W/16,C2), (B,H/32,W/32,C3).
 * feat16: NHWC fp32 (B,H/16,W/16,C2) = seq_dict["feat"].  H, W multiples of 32. */
int uni_backbone_fpn(uni_ctx* ctx, const float* img, int B, int H, int W, float* fpn0, float* fpn1, float* fpn2,
                     float* feat16, uni_stream_t stream);
/* feat_*: NHWC fp32 (B,h,w,C2); pos_*: NHWC fp32 (h,w,256) shared by the batch; out_*: NHWC fp32 (B,h,w,256). */
int uni_interaction(uni_ctx* ctx, const float* feat_ref, const float* pos_ref, const float* feat_cur,
                    const float* pos_cur, int B, int h, int w, float* out_ref, float* out_cur, uni_stream_t stream);
/* feat: NHWC fp32 (B,h,w,256) -> embed: NHWC fp32 (B,2h,2w,embed_dim). */
int uni_upsample(uni_ctx* ctx, const float* feat, int B, int h, int w, float* embed, uni_stream_t stream);
/* fpn*: as produced by uni_backbone_fpn for B (H,W) images; prior*: (B,H/8*W/8), (B,H/16*W/16), (B,H/32*W/32) fp32
 * (one prior per image and level).  mode bit 0: 0 = "sot", 1 = "mot"; bit 1 (value 2): raw rows, i.e. the reference's
 * decode_in_inference = False (unicorn_head.py:436-439, tools/export_torchscript.py:66; not for mask models).
 * out: (B, A, 5+nc) fp32 decoded, A = sum of level
 * sizes, nc = 1 (sot) or num_classes (mot).  Mask models additionally fill dyn_params (B,A,169), mask_feats
 * (B,H/8,W/8,8) NHWC and up_masks (B,H/8,W/8,9*up_rate^2) NHWC (pass NULL for box-only models). */
int uni_head(uni_ctx* ctx, const float* fpn0, const float* fpn1, const float* fpn2, const float* prior8,
             const float* prior16, const float* prior32, int B, int H, int W, int mode, float* out, float* dyn_params,
             float* mask_feats, float* up_masks, uni_stream_t stream);
/* Object-batched head (VOS, external/lib/test/tracker/unicorn_vos.py:178-200 runs the head once per object on the SAME
 * FPN maps): ONE image, K prior sets prior*: (K, H/s*W/s).  The prior enters only at x = stem(fpn) + prior*beta
 * (unicorn_head.py:272-277), so FPN casts, stem convs and the mask branch run once and the rest over K samples.
 * out (K, A, 5+nc), dyn_params (K, A, 169); mask_feats / up_masks are those of the ONE image (1, ...). */
int uni_head_objects(uni_ctx* ctx, const float* fpn0, const float* fpn1, const float* fpn2, const float* prior8,
                     const float* prior16, const float* prior32, int K, int H, int W, int mode, float* out,
                     float* dyn_params, float* mask_feats, float* up_masks, uni_stream_t stream);
int uni_pos_embed(uni_ctx* ctx, int h, int w, float* out_nhwc, uni_stream_t stream);

/* ---- context-free operators ------------------------------------------------------------------------- */
/* value [N,S,M,D] fp32, spatial_shapes [L,2] int64 HOST, level_start_index [L] int64 HOST,
 * sampling_loc [N,Lq,M,L,P,2], attn_weight [N,Lq,M,L,P] -> out [N,Lq,M*D]. */
int uni_msda_fwd(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                 const float* sampling_loc, const float* attn_weight, float* out, int N, int S, int M, int D, int Lq,
                 int L, int P, uni_stream_t stream);
/* e_ref [R,128], e_cur [Q,128] fp32 row-major (NHWC embedding maps), values [K,R] -> out [K,Q].
 * precision 0 = exact fp32 MFMA, 1 = fp32-equivalent bf16x3 split (6 bf16 MFMAs per product), 2 = fp32-equivalent f16x2 split
 * (3 f16 MFMAs per product; operands must lie inside the f16 range, |x| < 65504 / log2 e), 3 = the reference DRIVER's arithmetic
 * class (unicorn_sot.py:95-100 casts keys, queries, values to fp16): f16-rounded operands, one MFMA per product, f16-rounded scores
 * (not the `.half()` of the normalised softmax).  workspace: device scratch of >= uni_corr_workspace_bytes bytes. */
size_t uni_corr_workspace_bytes(int R, int Q, int K);
int uni_corr_softmax_pv(const float* e_ref, const float* e_cur, const float* values, float* out, int R, int Q, int D,
                        int K, int precision, void* workspace, size_t workspace_bytes, uni_stream_t stream);
/* The same for B frames in ONE launch (the time-batched SOT step, unicorn_sot.py:88-105 per frame): e_ref [B,R,128], e_cur [B,Q,128],
 * values [K,R] shared by the frames (values_per_frame = 0) or [B,K,R] -> out [B,K,Q].  workspace >= uni_corr_workspace_bytes_batched. */
size_t uni_corr_workspace_bytes_batched(int B, int R, int Q, int K);
int uni_corr_softmax_pv_batched(const float* e_ref, const float* e_cur, const float* values, float* out, int B, int R, int Q, int D,
                                int K, int values_per_frame, int precision, void* workspace, size_t workspace_bytes,
                                uni_stream_t stream);
int uni_prior_pyramid(const float* p8, float* p16, float* p32, int K, int H8, int W8, uni_stream_t stream);
int uni_label_map_s8(const float* box_xyxy_dev, float* out, int H, int W, uni_stream_t stream);
int uni_sample_embeddings(const float* embed_nhwc, int H8, int W8, int C, const float* boxes_xyxy, int ld_boxes, int n,
                          float stride, float* out, uni_stream_t stream);
/* mask_feats (H8,W8,8), up_masks (H8,W8,9*r*r) NHWC fp32; params (n,169) row stride ldp; inst_loc (n,2);
 * inst_lvl (n) int32 -> out (n, d_rate*r*H8, d_rate*r*W8) sigmoid scores.  workspace >= n*H8*W8*(1+r*r)*4 bytes */
int uni_condinst_masks(const float* mask_feats, const float* up_masks, const float* params, int ldp,
                       const float* inst_loc, const int32_t* inst_lvl, int n, int H8, int W8, int up_rate, int d_rate,
                       float* out, void* workspace, size_t workspace_bytes, uni_stream_t stream);

/* Input letterbox on the device (row 0 / N1): PreprocessorX.process (external/lib/test/tracker/unicorn_sot.py:111-123,
 * swap_rb = 1) and preproc (unicorn/data/data_augment.py:194-214, swap_rb = 0).  img_hwc: (h, w, 3) uint8 DEVICE buffer;
 * out_chw: (3, H, W) fp32 = cv2.resize(INTER_LINEAR, 8-bit fixed point) to (int(w r), int(h r)), r = min(H/h, W/w),
 * top-left aligned, padded with 114.  *r_out (host, optional) receives r. */
int uni_letterbox(const uint8_t* img_hwc, int h, int w, int swap_rb, int H, int W, float* out_chw, double* r_out,
                  uni_stream_t stream);

/* UnicornHead.decode_outputs (unicorn_head.py:467-482), in place: outputs (B, A, nch) raw [dx,dy,log w,log h,...] over the three
 * levels (strides 8/16/32 of an HxW input, level-major like the head's concat) -> xy = (xy + grid) * stride, wh = exp(wh) * stride.
 * (uni_head already returns decoded outputs; this entry serves callers that keep decode_in_inference=False.) */
int uni_decode_outputs(float* outputs, int B, int H, int W, int nch, uni_stream_t stream);
/* torchvision.ops.nms on caller boxes (unicorn/utils/boxes.py:58-64): boxes (n,4) xyxy, scores (n) -> keep_idx (n) int32 in
 * descending-score order, *n_out (device int32).  workspace >= uni_nms_workspace_bytes(n). */
size_t uni_nms_workspace_bytes(int n);
int uni_nms(const float* boxes_xyxy, const float* scores, int n, float iou_thr, int32_t* keep_idx, int32_t* n_out, void* workspace,
            size_t workspace_bytes, uni_stream_t stream);

/* Detection post-processing of ONE image on the device (row N1): unicorn/utils/boxes.py:33-77 `postprocess`
 * (+ torchvision.ops.nms / batched_nms semantics).  pred: (A, ld >= 5+num_classes) decoded [cx,cy,w,h,obj,cls...] fp32,
 * converted to corners IN PLACE like the reference (:36-39).  Survivors, in descending obj*cls order:
 * det_out (max_det, 7) rows [x1,y1,x2,y2,obj,cls_conf,cls], keep_idx (max_det) anchor indices, *n_out = row count
 * (device int32, clamped to max_det).  flags: bit 0 = class-agnostic NMS, bit 1 = boxes are already corners (plain
 * torchvision-style nms on caller boxes).  workspace >= uni_postprocess_workspace_bytes(A) device bytes. */
size_t uni_postprocess_workspace_bytes(int A);
int uni_postprocess(float* pred, int A, int ld, int num_classes, float conf_thre, float nms_thre, int flags,
                    int max_det, float* det_out, int32_t* keep_idx, int32_t* n_out, void* workspace, size_t workspace_bytes,
                    uni_stream_t stream);

/* ---- mask post-processing of the VOS / MOTS drivers -------------------------------------------- */
/* masks (N,Hn,Wn) fp32 at network resolution -> F.interpolate(scale_factor=1/r, bilinear, align_corners=False)[:, :H, :W]
 * pasted into zero (N,H,W) maps: out_prob fp32 (external/lib/test/tracker/unicorn_vos.py:146-150) and / or
 * out_bin = prob > thr as bytes (unicorn/evaluators/mot_evaluator.py:804-805).  Either output may be NULL. */
int uni_mask_resize(const float* masks, int N, int Hn, int Wn, double r, int H, int W, float thr, float* out_prob,
                    uint8_t* out_bin, uni_stream_t stream);
/* postprocess_inst's mask half + the MOTS threshold in ONE call, for callers that never look at the network-size maps
 * (unicorn/utils/boxes.py:138-146 -> unicorn/models/condinst/dynamic_mask_head.py:159-225 -> unicorn/evaluators/mot_evaluator.py:804-805;
 * the VOS paste of external/lib/test/tracker/unicorn_vos.py:141-152 with out_prob): arguments of uni_condinst_masks, then
 * F.interpolate(scale_factor=1/r, bilinear)[:, :H, :W] pasted into zero (n,H,W) maps as fp32 (out_prob) and / or `> thr` bytes (out_bin);
 * either may be NULL.  The (n, d_rate*r*H8, d_rate*r*W8) fp32 maps are never written; results are bit-identical to
 * uni_condinst_masks followed by uni_mask_resize.  workspace >= n*H8*W8*(1+r*r)*4 bytes. */
int uni_condinst_masks_u8(const float* mask_feats, const float* up_masks, const float* params, int ldp, const float* inst_loc,
                          const int32_t* inst_lvl, int n, int H8, int W8, int up_rate, int d_rate, double r, int H, int W, float thr,
                          float* out_prob, uint8_t* out_bin, void* workspace, size_t workspace_bytes, uni_stream_t stream);
/* Soft aggregation of unicorn_vos.py:99-120 fused with that resize: probs (K1,Hn,Wn) of the tracked objects (ids prob_ids, in
 * cur_obj_ids order), init_masks (K2,H,W) {0,1} of objects introduced in this frame (ids init_ids); background =
 * prod(1 - p), argmax over [background, ids] (numpy first-maximum rule) -> out (H,W) uint8 id map. */
int uni_vos_merge(const float* probs, const int32_t* prob_ids, int K1, int Hn, int Wn, double r, const uint8_t* init_masks,
                  const int32_t* init_ids, int K2, int H, int W, uint8_t* out, uni_stream_t stream);
/* mot_evaluator.py:860-865: masks (N,H,W) {0,1} in track order -> a pixel stays with the first mask that claims it. */
int uni_mots_overlap_free(const uint8_t* masks, int N, int H, int W, uint8_t* out, uni_stream_t stream);
/* pycocotools rleEncode + rleToString of np.asfortranarray(mask) (mot_evaluator.py:889-892): out_chars (N,max_chars) bytes,
 * out_len (N) string lengths (-1: more than max_runs runs or max_chars chars, retry with larger bounds); optional
 * counts (N,max_runs+1) uint32 run lengths and n_runs (N). */
size_t uni_rle_workspace_bytes(int N, int H, int W, int max_runs);
int uni_rle_encode(const uint8_t* masks, int N, int H, int W, int max_runs, int max_chars, uint8_t* out_chars,
                   int32_t* out_len, uint32_t* counts, int32_t* n_runs, void* workspace, size_t workspace_bytes,
                   uni_stream_t stream);

/* ---- low-level building blocks (exported for kernel parity tests) --------------------------------------- */
/* out[m][n] = act(sum_k A[m][k] W[n][k] + bias[n]) (+res).  A: bf16 NHWC map (Hin,Win,Cin) row stride lda.
 * w_packed: [roundup(N,256)][roundup(K,64)] bf16, K order (ky,kx,c) (see uni_pack_weight). */
int uni_pack_weight(const float* w_oihw_host, int N, int Cin, int KH, int KW, uint16_t* out_host_bf16);
int uni_gemm_bf16(const uint16_t* A, int lda, const uint16_t* w_packed, int M, int N, int Hin, int Win, int Cin, int KH,
                  int KW, int stride, int pad, const float* bias, int act, const float* residual, int ldr, float* outF,
                  int ldf, uint16_t* outB, int ldb, double* gn_stats, int cpg, int force_cfg, uni_stream_t stream);
int uni_cast_bf16(const float* x, int ldx, uint16_t* out, int ldo, int M, int C, uni_stream_t stream);
/* The same three for the "f16x2" operand format (precision 2: every value split into hi + lo f16 halves, per 8 channels a
 * 32-byte group [8 x hi][8 x lo]; fp32-equivalent contraction as hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16).
 * Buffers are 4 bytes per element.  uni_pack_weight_h2 returns the power-of-two scale the packed tensor carries in
 * *wscale_out (pass it to uni_gemm_h2, which multiplies the accumulator by it). */
int uni_pack_weight_h2(const float* w_oihw_host, int N, int Cin, int KH, int KW, void* out_host, float* wscale_out);
int uni_gemm_h2(const void* A, int lda, const void* w_packed, float wscale, int M, int N, int Hin, int Win, int Cin, int KH,
                int KW, int stride, int pad, const float* bias, int act, const float* residual, int ldr, float* outF,
                int ldf, void* outB, int ldb, double* gn_stats, int cpg, int force_cfg, uni_stream_t stream);
int uni_cast_h2(const float* x, int ldx, void* out, int ldo, int M, int C, uni_stream_t stream);
/* Fused ConvNeXt MLP of the narrow stages (unicorn/models/backbone/convnext.py:47-54; C in {96, 192, 256}):
 *     out[m][:] = residual[m][:] + b2 + diag(gamma) W2 . GELU(W1 . a[m][:] + b1)
 * in one launch with the 4C hidden units kept in registers (csrc/mlp_fused.hip).  uni_mlp_pack lays the nn.Linear weights
 * w1 [4C][C], w2 [C][4C] (host, fp32; gamma [C] or NULL folded into the rows of w2) out as the f16x2 weight stream the kernel
 * consumes (uni_mlp_blob_bytes(C) bytes, host; copy it to the device) and returns the two accumulator factors.  a: f16x2 operand
 * rows (uni_cast_h2 / uni_dwconv7_ln output), b2 must already carry gamma; out may alias residual; out_h2 (optional) receives
 * an f16x2 copy of the result.  layout 0: 4 waves x 32 rows per 128-row tile, one wave per SIMD (C = 96, 192, 256); layout 1: 8 waves x
 * 16 rows, two waves per SIMD, v_mfma_f32_16x16x32_f16 (C = 192, 256; what the engine uses).  The blob is layout specific. */
size_t uni_mlp_blob_bytes(int C);
int uni_mlp_pack(const float* w1_host, const float* w2_host, const float* gamma_host, int C, int layout, void* blob_host,
                 float* ws1_out, float* ws2_out);
int uni_mlp_fused(const void* a_h2, int lda, const void* blob_dev, const float* b1, const float* b2, float ws1, float ws2,
                  const float* residual, int ldr, float* out, int ldo, void* out_h2, int ldb, int M, int C, int layout, int dbg,
                  uni_stream_t stream);
int uni_layernorm(const float* x, int ldx, const float* gamma, const float* beta, float eps, int M, int C, float* outF,
                  uint16_t* outB, uni_stream_t stream);
int uni_dwconv7_ln(const float* x_nhwc, const float* w49c, const float* bias, const float* gamma, const float* beta,
                   float eps, int H, int W, int C, uint16_t* out_bf16, uni_stream_t stream);
/* The same for B stacked (H,W,C) maps and any operand format (what the engine calls per ConvNeXt block): fmt 0 = bf16 rows, 1 = fp32
 * rows, 2 = f16x2 rows ([8 hi][8 lo] groups).  convnext.py:30-33,47-49. */
int uni_dwconv7_ln_ex(const float* x_nhwc, const float* w49c, const float* bias, const float* gamma, const float* beta, float eps,
                      int B, int H, int W, int C, void* out, int fmt, uni_stream_t stream);
/* The engine's sampler of the ref <-> cur interaction (csrc/msda.hip msda_wave_kernel: one wave per (token, head), shuffle reductions):
 * Unicorn's fixed geometry -- 8 heads x 32 channels, 2 levels = reference / current frame of identical (h, w), 4 points -- with
 * MSDeformAttn.forward's softmax over the 8 logits and loc = ref + off / (W, H) fused in (ms_deform_attn.py:98-105,
 * deformable_transformer.py:141-153).  value (B, 2 h w, 256) fp32; offaw (B * 2 h w, ldo >= 192): 128 sampling offsets
 * (head, level, point, xy) then 64 attention logits (head, level, point) -> out (B * 2 h w, 256) fp32. */
int uni_msda_tokens(const float* value, const float* offaw, int ldo, int B, int h, int w, float* out, uni_stream_t stream);
int uni_groupnorm_act(const float* x, const double* stats, const float* gamma, const float* beta, float eps, int M,
                      int C, int G, int act, float* outF, uint16_t* outB, uni_stream_t stream);
int uni_stem(const float* img, int H, int W, const float* w48c, const float* bias, const float* gamma,
             const float* beta, int C, float* out_nhwc, uni_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
