// Internal launcher interface between the HIP kernels (*.hip) and the engine / C-ABI (engine.hip, api.hip).
#pragma once
#include "common.h"

// ---------------------------------------------------------------- gemm.hip
struct GemmArgs {
    const bf16* A = nullptr; int lda = 0;     // NHWC bf16 activations, lda = elements per pixel row
    const bf16* W = nullptr;                  // packed [Npad][Kpad] bf16
    int M = 0, N = 0, K = 0, Kpad = 0;
    // conv geometry (1x1/s1/p0 => plain GEMM)
    int Hin = 0, Win = 0, Cin = 0, KH = 1, KW = 1, stride = 1, pad = 0, Wout = 0;
    int Mper = 0;                             // output pixels per sample (M = B * Mper); GN stats slot b lives at stats + 64*b
    int out_hw = 0, out_stride = 0, out_off = 0;   // optional outF row remap: (row/out_hw)*out_stride + out_off + row%out_hw
    // epilogue
    const float* bias = nullptr;              // [N]
    int act = ACT_NONE; int act_col0 = 0;     // activation on columns >= act_col0
    const float* res = nullptr; int ldr = 0;  // fp32 residual added after the activation
    float* outF = nullptr; int ldf = 0;
    bf16* outB = nullptr; int ldb = 0;
    double* stats = nullptr; int cpg = 0;     // GroupNorm group sums [G][2] of (acc+bias), cpg = channels/group
    int force_cfg = 0;                        // 0 = heuristic, else 22 / 12 / 21 / 11
    int b32 = 0;                              // operand format (ActFmt): FMT_BF16, FMT_F32 (A, W, outB fp32; v_mfma_f32_32x32x2_f32) or FMT_H2 (split f16, gemm_h2.hip)
    float wscale = 1.f;                       // FMT_H2: the packed weights carry a power-of-two scale; the accumulator is multiplied by wscale
    // FMT_H2, single-frame problems (few tiles): splitk > 1 cuts the K loop into `splitk` ranges (gridDim.y); every range stores its partial
    // tile into slab [splitk][M][N] fp32 and a reduce kernel (launched by launch_gemm_h2) sums the ranges and applies bias / residual /
    // GroupNorm sums -> outF.  No activation, no operand-format output in this mode.
    int splitk = 0; float* slab = nullptr;
    int dbg = 0;                              // ablation switches for tools/gemm_bench.py (1 = no DMA after tile 0, 2 = no MFMA)
    int epi = 0;                              // set by launch_gemm: 1 = LDS-staged, row-coalesced epilogue stores
};
int launch_gemm(const GemmArgs& a, hipStream_t s);
int launch_gemm_h2(const GemmArgs& a, hipStream_t s);     // gemm_h2.hip (reached through launch_gemm when a.b32 == FMT_H2)
uint16_t f32_to_f16_host(float f);
float f16_to_f32_host(uint16_t h);
float h2_weight_scale(float maxabs);
void pack_weight_h2_host(const float* w, int N, int Cin, int KH, int KW, const float* row_scale, float scale, uint16_t* out,
                         int Npad, int Kpad);
int launch_gemm_p44(const GemmArgs& a, hipStream_t s);    // gemm_p44.hip: persistent 256x256 tiles (bf16), next tile prefetched before the drain
bool gemm_p44_supported(const GemmArgs& a);
// deep-pipeline 4-wave tiles for launches with <= ~1 block per CU (gemm_h2d.hip); cfg 322 / 323 / 321 / 312 / 311
bool gemm_h2d_has_cfg(int cfg);
bool gemm_h2d_supported(const GemmArgs& a, int cfg);
int gemm_h2d_choice(const GemmArgs& a);      // 0 or the deep tile configuration the launcher takes for this problem
int launch_gemm_h2d(const GemmArgs& a, int cfg, bool conv, hipStream_t s);
GemmArgs gemm_splitk_partial_args(const GemmArgs& a);    // the launch that fills the slab: fp32 partial tiles only (no bias / residual / statistics / remap)
int launch_splitk_reduce(const GemmArgs& a, hipStream_t s);
int launch_gemm_h2q(const GemmArgs& a, hipStream_t s);    // gemm_h2q.hip: persistent 256x256, two wave groups ping-pong MFMA / LDS phases, counted-vmcnt DMA stream
bool gemm_h2q_supported(const GemmArgs& a);

// ---------------------------------------------------------------- mlp_fused.hip
// Fused ConvNeXt MLP (pwconv1 -> GELU -> pwconv2 (gamma folded) -> + residual), f16x2 operands, hidden kept in registers
struct MlpArgs {
    const void* A = nullptr; int lda = 0;     // [M][C] f16x2 operand rows (dwconv7 + LayerNorm output)
    const void* blob = nullptr;               // weight stream of mlp_pack_host (mlp_blob_bytes(C) bytes)
    const float* b1 = nullptr;                // [4C]
    const float* b2 = nullptr;                // [C], layer scale folded in
    float ws1 = 1.f, ws2 = 1.f;               // accumulator factors (1 / power-of-two weight scale)
    const float* res = nullptr; int ldr = 0;  // fp32 residual rows (may alias out)
    float* out = nullptr; int ldo = 0;        // fp32 [M][C]
    void* outB = nullptr; int ldb = 0;        // optional f16x2 copy of the result
    int M = 0, C = 0;
    int layout = 0;                           // 0: 32-row waves (4 per block, blob of mlp_pack_host); 1: 16-row waves (8 per block, mlp_pack16_host)
    int dbg = 0;                              // ablation switches (tools/mlp_bench.py, C = 192): 1 no DMA, 2 no MFMA, 16 no GELU arithmetic, 8 (any C) every block starts its weight stream at hidden block 0
};
int launch_mlp_fused(const MlpArgs& a, hipStream_t s);
bool mlp_fused_supported(int C);
size_t mlp_blob_bytes(int C);
void mlp_pack_host(const float* w1, const float* w2, const float* gamma, int C, uint16_t* out, float* ws1, float* ws2);
void mlp_pack16_host(const float* w1, const float* w2, const float* gamma, int C, uint16_t* out, float* ws1, float* ws2);
bool mlp_fused16_supported(int C);

// ---------------------------------------------------------------- norm.hip
// Row LayerNorm over C (biased var, eps inside sqrt): fp32 [M][ldx] -> bf16 and/or fp32.
struct LnArgs {
    const float* x = nullptr; int ldx = 0;
    const float* gamma = nullptr; const float* beta = nullptr; float eps = 1e-6f;
    int M = 0, C = 0;
    float* outF = nullptr; int ldf = 0;
    bf16* outB = nullptr; int ldb = 0;
    // optional PixelShuffle(2) scatter of the bf16 output (unicorn.py:41): row=(y,x) of an (h,w) map,
    // channel c -> pixel (2y+dy, 2x+dx), channel c/4 of a (2h,2w,C/4) map
    int ps_h = 0, ps_w = 0;
    int b32 = 0;                              // outB holds fp32 instead of bf16
    // optional fp32 row remap of the interaction stage: rows are tokens [B][2 frames][pair_hw]; frame 0 rows go to outF, frame 1 rows to
    // outF2, both as [B][pair_hw] maps (one launch instead of 2 B)
    float* outF2 = nullptr; int pair_hw = 0;
};
int launch_layernorm(const LnArgs& a, hipStream_t s);

// GroupNorm apply from group sums + activation (+ prior fusion), fp32 [M][C] raw -> bf16 / fp32 (+2x nearest copy)
struct GnApplyArgs {
    const float* x = nullptr; int ldx = 0;
    const double* stats = nullptr;            // [G][2] sum, sumsq
    const float* gamma = nullptr; const float* beta = nullptr; float eps = 1e-3f;
    int M = 0, C = 0, G = 16, act = ACT_SILU;    // M = rows PER SAMPLE
    int B = 1;                                   // samples; x/out rows are [B*M], stats slot b at stats + 64*b
    const float* prior = nullptr; const float* prior_beta = nullptr;   // y += prior[m] * prior_beta[c]
    float* outF = nullptr; int ldf = 0;
    bf16* outB = nullptr; int ldb = 0;
    bf16* outUp = nullptr; int ldu = 0; int W = 0;   // 2x nearest upsampled copy into a (2H,2W) map
    int b32 = 0;
};
int launch_gn_apply(const GnApplyArgs& a, hipStream_t s);

// depthwise 7x7 (+bias) + LayerNorm(C): fp32 NHWC -> bf16 [M][C]
struct DwLnArgs {
    const float* x = nullptr;                 // [H][W][C]
    const float* w = nullptr;                 // [49][C]
    const float* bias = nullptr; const float* gamma = nullptr; const float* beta = nullptr;
    float eps = 1e-6f;
    int H = 0, W = 0, C = 0, B = 1;            // B images of (H,W,C) stacked
    bf16* out = nullptr;
    int b32 = 0;
};
int launch_dwconv7_ln(const DwLnArgs& a, hipStream_t s);

// stem: conv4x4/s4 (3->C) + bias + LN_cf: NCHW fp32 image -> fp32 NHWC
struct StemArgs {
    const float* img = nullptr; int H = 0, W = 0, B = 1;   // (B,3,H,W)
    const float* w = nullptr;                       // [48][C]  (k = c*16 + ky*4 + kx)
    const float* bias = nullptr; const float* gamma = nullptr; const float* beta = nullptr;
    int C = 0; float* out = nullptr;                // [H/4][W/4][C]
};
int launch_stem(const StemArgs& a, hipStream_t s);

// ---------------------------------------------------------------- msda.hip
// reference-compatible op (ops/src/vision.cpp:13-16): value [N,S,M,D], loc [N,Lq,M,L,P,2], attn [N,Lq,M,L,P]
int launch_msda(const float* value, const int64_t* shapes_host, const int64_t* lsi_host, const float* loc,
                const float* attn, float* out, int N, int S, int M, int D, int Lq, int L, int P, hipStream_t s);
// fused engine variant: raw offsets / attention logits -> softmax, loc = ref + off/(W,H), sample, bf16 out
struct MsdaFusedArgs {
    const float* value = nullptr;             // [2*hw][256] fp32
    const float* offaw = nullptr; int ldo = 0;   // [Lq][192]: 128 offsets (m,l,p,xy) | 64 logits (m,l,p)
    int h = 0, w = 0;                         // both levels (h,w); Lq = 2*h*w
    bf16* out = nullptr;                      // [Lq][256]
    int b32 = 0;
    int B = 1;                                // batch: tokens [B][2][hw]
};
int launch_msda_fused(const MsdaFusedArgs& a, hipStream_t s);

// ---------------------------------------------------------------- corr.hip
// out[k][q] = sum_r V[k][r] softmax_r(<Eref[r], Ecur[q]>);  Eref [R][D], Ecur [Q][D] fp32 row-major, V [K][R]
int launch_corr(const float* eref, const float* ecur, const float* v, float* out, int R, int Q, int D, int K,
                int precision, void* workspace, size_t ws_bytes, hipStream_t s);
size_t corr_workspace_bytes(int R, int Q, int K);
int launch_corr_batched(const float* eref, const float* ecur, const float* v, float* out, int B, int R, int Q, int D, int K,
                        int values_per_frame, int precision, void* workspace, size_t ws_bytes, hipStream_t s);
size_t corr_workspace_bytes_batched(int B, int R, int Q, int K);

// ---------------------------------------------------------------- misc.hip
int launch_cast_operand(const float* x, int ldx, bf16* out, int ldo, int M, int C, hipStream_t s, int b32 = 0);
// x0 / x1 [B][hw][C] fp32 (two frames) -> out [B][2][hw][C] in the operand format (the token layout of the interaction stage), one launch
int launch_cast_operand_pair(const float* x0, const float* x1, bf16* out, int hw, int C, int B, hipStream_t s, int b32);
int launch_pixel_shuffle_bf16(const float* x, bf16* out, int h, int w, int C, hipStream_t s, int b32 = 0, int B = 1);
int launch_prior_pyramid(const float* p8, float* p16, float* p32, int K, int H8, int W8, hipStream_t s);
int launch_decode(const float* raw, float* out, int A0, int W0, int A1, int W1, int A2, int W2, int nch, hipStream_t s, int B = 1);
int launch_add_aligned_bilinear(const float* src, int h, int w, int C, int factor, float* dst, hipStream_t s, int B = 1);
int launch_pos_embed(const float* row, const float* col, int sz, int nf, float* out, int h, int w, hipStream_t s);
int launch_sample_embed(const float* emb, int H, int W, int C, const float* boxes, int ldbox, int n, float stride,
                        float* out, hipStream_t s);
struct CondInstArgs {
    const float* mask_feats = nullptr;        // [H][W][8] fp32
    const float* up_masks = nullptr;          // [H][W][9*r*r] fp32
    const float* params = nullptr; int ldp = 0;   // [n][169]
    const float* inst_loc = nullptr;          // [n][2]
    const int* inst_lvl = nullptr;            // [n]
    int n = 0, H = 0, W = 0, r = 4, d_rate = 2;
    float* logits_ws = nullptr;               // [n][H][W] workspace
    float* coarse_ws = nullptr;               // [n][rH][rW] workspace (sigmoid scores at 1/d_rate res)
    float* out = nullptr;                     // [n][d_rate*r*H][d_rate*r*W]
};
int launch_condinst(const CondInstArgs& a, hipStream_t s);
int launch_label_map_s8(const float* box_xyxy, float* out, int H, int W, hipStream_t s);
// post.hip: utils/boxes.py:33-77 on the device (corners in place, conf filter, (batched) NMS, sorted survivor rows)
int launch_letterbox(const unsigned char* img, int h, int w, int swap_rb, int H, int W, float* out, double* r_out, hipStream_t s);
size_t postprocess_workspace_bytes(int A);
size_t nms_workspace_bytes(int n);
int launch_nms(const float* boxes, const float* scores, int n, float iou_thr, int32_t* keep_idx, int32_t* n_out, void* ws,
               size_t ws_bytes, hipStream_t s);
int launch_postprocess(float* pred, int A, int ld, int num_classes, float conf_thre, float nms_thre, int flags,
                       int max_det, float* det_out, int32_t* keep_idx, int32_t* n_out, void* ws, size_t ws_bytes, hipStream_t s);
// mask_post.hip: mask post-processing of the VOS / MOTS drivers (row N1)
// coarse (N, h, w) CondInst sigmoid scores at 1 / d_rate of the network resolution -> aligned_bilinear(x f) -> resize by 1 / r -> fp32 and / or
// `> thr` bytes at (H, W), without the (N, f h, f w) intermediate (bit-identical to launch_condinst's last pass + launch_mask_resize)
int launch_condinst_resize(const float* coarse, int N, int h, int w, int f, float rscale, int ho, int wo, int H, int W, float thr, float* outF,
                           unsigned char* outU, hipStream_t s);
int launch_mask_resize(const float* masks, int N, int Hn, int Wn, float rscale, int ho, int wo, int H, int W, float thr, float* outF,
                       unsigned char* outU, hipStream_t s);
int launch_vos_merge(const float* probs, const int* prob_ids, int K1, int Hn, int Wn, float rscale, int ho, int wo,
                     const unsigned char* init_masks, const int* init_ids, int K2, int H, int W, unsigned char* out, hipStream_t s);
int launch_overlap_free(const unsigned char* in, int N, int H, int W, unsigned char* out, hipStream_t s);
size_t rle_workspace_bytes(int N, int H, int W, int max_runs);
int launch_rle_encode(const unsigned char* masks, int N, int H, int W, int max_runs, int max_chars, unsigned char* out_chars,
                      int* out_len, unsigned* counts, int* n_runs, void* ws, size_t ws_bytes, hipStream_t s);
int launch_add_pos_bf16(const float* src, const float* pos0, const float* pos1, const float* lvl, bf16* out, int hw,
                        int C, hipStream_t s, int b32 = 0, int B = 1);
