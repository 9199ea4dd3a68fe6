// uni_ctx: re-packed weights + workspace of one Unicorn model instance (see engine.hip).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "../../include/unicorn_hip.h"
#include "kernels.h"

#define UNI_SLAB_BYTES ((size_t)32 << 20)   // budget of one split-K partial-tile slab (engine.hip choose_splitk); every concurrently running level owns one
#define UNI_STATS_SLOTS 4096     // GroupNorm statistics slots per stage call: one per (GroupNorm, sample); backbone_fpn has 36 GNs, the mask head 31, B <= 64

struct HostParam { std::vector<int64_t> shape; std::vector<float> data; };
struct PConv { bf16* W = nullptr; float* bias = nullptr; int N = 0, K = 0, Kpad = 0, KH = 1, KW = 1, Cin = 0, b32 = 0; float wscale = 1.f; };
struct PAffine { float* g = nullptr; float* b = nullptr; };
struct PBlock { float* dw_w = nullptr; float* dw_b = nullptr; PAffine ln; PConv pw1, pw2; int C = 0;
                void* mlp_blob = nullptr; float mlp_ws1 = 1.f, mlp_ws2 = 1.f; int mlp_layout = 0; };   // mlp_blob: weight stream of the fused MLP kernel (mlp_fused.hip), f16x2 mode and C in {96, 192, 256}   // ln_folded: pw1 carries the LN gamma (weights) / beta (bias), colsum = row sums of the packed weights
struct PBaseConv { PConv conv; PAffine gn; int k = 1, stride = 1; };
struct PCsp { PConv c12; PAffine gn12; PBaseConv m1[3], m2[3], c3; int cin = 0, cout = 0, h = 0; };

struct ProfRec { hipEvent_t a, b; double work; int cls; int M = 0, N = 0, K = 0, conv = 0; };

struct uni_ctx {
    int device = 0;
    uni_model_cfg cfg{};
    std::map<std::string, HostParam> host;
    std::vector<std::string> missing;
    std::vector<float> zeros;
    std::vector<void*> dev_allocs;
    bool finalized = false, failed = false;
    int b32 = 0;   // precision mode = operand format (ActFmt): 0 = bf16 MFMA operands, 1 = exact fp32 (v_mfma_f32_32x32x2_f32), 2 = split f16 ("f16x2", fp32-equivalent)
    // ConvNeXt
    float* stem_w = nullptr; float* stem_b = nullptr; PAffine stem_ln;
    PAffine ds_ln[4]; PConv ds_conv[4];
    std::vector<PBlock> blocks[4];
    PAffine out_norm[4];
    // PAFPN
    PBaseConv lateral0, reduce1, bu2, bu1;
    PCsp c3p4, c3p3, c3n3, c3n4;
    // head
    PBaseConv stems[3]; float* beta[3] = {nullptr, nullptr, nullptr};
    std::vector<PBlock> att[3];
    PConv tower0[3]; PAffine tower0_gn[3];
    PBaseConv cls_convs[3][4], reg_convs[3][4];
    PConv cls_pred[3], cls_pred_sot[3], regobj[3], regobj_sot[3], controllers[3];
    PConv refine[3]; PAffine refine_gn[3]; PConv mtower[4]; PAffine mtower_gn[4]; PConv mtower_out, upm0, upm1;
    // interaction / embedding
    PConv bott; PAffine bott_gn; PConv value_proj, offaw, output_proj, lin1, lin2; PAffine norm1, norm2;
    float* level_embed = nullptr; PConv up1, up3; float* pos_row = nullptr; float* pos_col = nullptr;
    // scratch
    char* ws = nullptr; size_t ws_cap = 0, ws_off = 0; bool ws_overflow = false;
    double* stats = nullptr; int stats_slot = 0;
    int nb = 1;   // batch size of the stage call in flight
    bool prof_on = false; std::vector<ProfRec> recs; double prof_bytes = 0.0;
    bool check_sat = false; unsigned long long* sat_dev = nullptr;   // uni_ctx_set_check: [saturated f16x2 operands, operands scanned, buffers scanned]
    hipStream_t aux[3] = {nullptr, nullptr, nullptr}; hipEvent_t ev_fork = nullptr; hipEvent_t ev_join[3] = {nullptr, nullptr, nullptr};   // head: levels 16 / 32 and the mask branch run beside level 8
};

uint16_t f32_to_bf16_host(float f);
void pack_weight_host(const float* w, int N, int Cin, int KH, int KW, const float* row_scale, uint16_t* out, int Npad, int Kpad);
int engine_finalize(uni_ctx* c);
int engine_reserve(uni_ctx* c, int B, int H, int W);
int engine_prof_begin(uni_ctx* c);
int engine_prof_end(uni_ctx* c, double* out);
int engine_set_check(uni_ctx* c, int on);
int engine_stats(uni_ctx* c, long long* out4);
void engine_destroy(uni_ctx* c);
int engine_backbone_fpn(uni_ctx* c, const float* img, int B, int H, int W, float* fpn0, float* fpn1, float* fpn2, float* feat16, hipStream_t s);
int engine_interaction(uni_ctx* c, const float* feat_ref, const float* pos_ref, const float* feat_cur, const float* pos_cur,
                       int B, int h, int w, float* out_ref, float* out_cur, hipStream_t s);
int engine_upsample(uni_ctx* c, const float* feat, int B, int h, int w, float* embed, hipStream_t s);
int engine_pos_embed(uni_ctx* c, int h, int w, float* out, hipStream_t s);
int engine_head(uni_ctx* c, const float* fpn0, const float* fpn1, const float* fpn2, const float* prior8, const float* prior16,
                const float* prior32, int B, int K, int H, int W, int mode, float* out, float* dyn_params, float* mask_feats, float* up_masks,
                hipStream_t s);
