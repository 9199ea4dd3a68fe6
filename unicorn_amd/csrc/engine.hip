// Host-side engine: owns the re-packed weights + scratch workspace and sequences the HIP kernels for the
// four Unicorn.forward modes (backbone / interaction / upsample / head).  No torch, no Python: this is the
// C++ runtime behind the C-ABI in api.hip.  Every function only enqueues work on the caller's stream.
#include "engine.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

// ------------------------------------------------------------------------------------------------
// bf16 helpers (host)
// ------------------------------------------------------------------------------------------------
uint16_t f32_to_bf16_host(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                             // round to nearest even
    return (uint16_t)(u >> 16);
}

void pack_weight_host(const float* w, int N, int Cin, int KH, int KW, const float* row_scale, uint16_t* out, int Npad,
                      int Kpad) {
    // OIHW fp32 -> [Npad][Kpad] bf16 with k = (ky*KW + kx)*Cin + c, zero padded
    const int K = KH * KW * Cin;
    memset(out, 0, (size_t)Npad * Kpad * sizeof(uint16_t));
    for (int n = 0; n < N; ++n) {
        const float sc = row_scale ? row_scale[n] : 1.f;
        uint16_t* o = out + (size_t)n * Kpad;
        const float* wn = w + (size_t)n * K;
        for (int c = 0; c < Cin; ++c)
            for (int ky = 0; ky < KH; ++ky)
                for (int kx = 0; kx < KW; ++kx)
                    o[(ky * KW + kx) * Cin + c] = f32_to_bf16_host(sc * wn[(c * KH + ky) * KW + kx]);
    }
}

// ------------------------------------------------------------------------------------------------
// parameter registry / upload
// ------------------------------------------------------------------------------------------------
static const float* host_param(uni_ctx* c, const std::string& name, size_t n) {
    auto it = c->host.find(name);
    if (it == c->host.end() || it->second.data.size() != n) {
        c->missing.push_back(name);
        c->zeros.assign(std::max(c->zeros.size(), n), 0.f);
        return nullptr;
    }
    return it->second.data.data();
}

template <class T>
static T* dev_upload(uni_ctx* c, const T* h, size_t n) {
    T* d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) {
        uni_set_error("hipMalloc of %zu bytes failed", n * sizeof(T));
        c->failed = true;
        return nullptr;
    }
    c->dev_allocs.push_back(d);
    if (h) {
        if (hipMemcpy(d, h, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) c->failed = true;
    } else {
        if (hipMemset(d, 0, n * sizeof(T)) != hipSuccess) c->failed = true;
    }
    return d;
}

static float* up_vec(uni_ctx* c, const std::string& name, size_t n) { return dev_upload<float>(c, host_param(c, name, n), n); }

static PAffine up_affine(uni_ctx* c, const std::string& prefix, int C) {
    PAffine a;
    a.g = up_vec(c, prefix + "weight", C);
    a.b = up_vec(c, prefix + "bias", C);
    return a;
}

struct ConvSrc { std::string w, b; int N; const float* row_scale = nullptr; };

// one or several (row-concatenated) OIHW convs sharing (Cin,KH,KW) -> packed bf16 + fp32 bias
static PConv pack_convs(uni_ctx* c, const std::vector<ConvSrc>& srcs, int Cin, int KH, int KW) {
    PConv p;
    p.Cin = Cin; p.KH = KH; p.KW = KW;
    p.K = Cin * KH * KW;
    p.Kpad = cdiv(p.K, 64) * 64;
    int N = 0;
    bool any_bias = false;
    for (auto& s : srcs) { N += s.N; any_bias |= !s.b.empty(); }
    p.N = N;
    const int Npad = cdiv(N, 256) * 256;   // widest block tile is 256 output channels
    std::vector<uint16_t> packed((size_t)Npad * p.Kpad, 0);
    std::vector<float> bias(N, 0.f);
    int n0 = 0;
    for (auto& s : srcs) {
        const float* w = host_param(c, s.w, (size_t)s.N * p.K);
        if (w) {
            std::vector<uint16_t> tmp((size_t)cdiv(s.N, 1) * p.Kpad);
            pack_weight_host(w, s.N, Cin, KH, KW, s.row_scale, tmp.data(), s.N, p.Kpad);
            memcpy(packed.data() + (size_t)n0 * p.Kpad, tmp.data(), tmp.size() * sizeof(uint16_t));
        }
        if (!s.b.empty()) {
            const float* b = host_param(c, s.b, s.N);
            if (b)
                for (int i = 0; i < s.N; ++i) bias[n0 + i] = b[i] * (s.row_scale ? s.row_scale[i] : 1.f);
        }
        n0 += s.N;
    }
    p.b32 = c->b32;
    if (c->b32 == FMT_H2) {   // split-f16 mode: [Npad][Kpad] x (hi, lo) f16 in 32-byte groups of 8 k, one power-of-two scale per packed tensor
        float mx = 0.f;
        for (auto& s : srcs) {
            const float* w = host_param(c, s.w, (size_t)s.N * p.K);
            if (w)
                for (int n = 0; n < s.N; ++n) {
                    const float sc = s.row_scale ? fabsf(s.row_scale[n]) : 1.f;
                    for (int k = 0; k < p.K; ++k) mx = std::max(mx, sc * fabsf(w[(size_t)n * p.K + k]));
                }
        }
        const float scale = h2_weight_scale(mx);
        p.wscale = 1.f / scale;
        std::vector<uint16_t> ph((size_t)Npad * p.Kpad * 2, 0);
        int r0 = 0;
        for (auto& s : srcs) {
            const float* w = host_param(c, s.w, (size_t)s.N * p.K);
            if (w) pack_weight_h2_host(w, s.N, Cin, KH, KW, s.row_scale, scale, ph.data() + (size_t)r0 * p.Kpad * 2, s.N, p.Kpad);
            r0 += s.N;
        }
        p.W = reinterpret_cast<bf16*>(dev_upload<uint16_t>(c, ph.data(), ph.size()));
    } else if (c->b32 == FMT_F32) {   // exact-fp32 mode: same [Npad][Kpad] (ky,kx,c) layout, fp32 elements
        std::vector<float> pf((size_t)Npad * p.Kpad, 0.f);
        int r0 = 0;
        for (auto& s : srcs) {
            const float* w = host_param(c, s.w, (size_t)s.N * p.K);
            if (w)
                for (int n = 0; n < s.N; ++n) {
                    const float sc = s.row_scale ? s.row_scale[n] : 1.f;
                    for (int ci = 0; ci < Cin; ++ci)
                        for (int ky = 0; ky < KH; ++ky)
                            for (int kx = 0; kx < KW; ++kx)
                                pf[(size_t)(r0 + n) * p.Kpad + (ky * KW + kx) * Cin + ci] = sc * w[(size_t)n * p.K + (ci * KH + ky) * KW + kx];
                }
            r0 += s.N;
        }
        c->missing.resize(c->missing.size());   // (host_param already recorded misses above)
        p.W = reinterpret_cast<bf16*>(dev_upload<float>(c, pf.data(), pf.size()));
    } else {
        p.W = reinterpret_cast<bf16*>(dev_upload<uint16_t>(c, packed.data(), packed.size()));
    }
    p.bias = any_bias ? dev_upload<float>(c, bias.data(), N) : nullptr;
    return p;
}
static PConv pack_conv(uni_ctx* c, const std::string& w, const std::string& b, int N, int Cin, int KH = 1, int KW = 1,
                       const float* row_scale = nullptr) {
    return pack_convs(c, {ConvSrc{w, b, N, row_scale}}, Cin, KH, KW);
}

static PBlock pack_block(uni_ctx* c, const std::string& p, int C) {
    PBlock b;
    b.C = C;
    // depthwise (C,1,7,7) -> [49][C]
    const float* dw = host_param(c, p + "dwconv.weight", (size_t)C * 49);
    std::vector<float> t((size_t)49 * C, 0.f);
    if (dw)
        for (int ch = 0; ch < C; ++ch)
            for (int k = 0; k < 49; ++k) t[(size_t)k * C + ch] = dw[(size_t)ch * 49 + k];
    b.dw_w = dev_upload<float>(c, t.data(), t.size());
    b.dw_b = up_vec(c, p + "dwconv.bias", C);
    b.ln = up_affine(c, p + "norm.", C);
    const float* w1 = host_param(c, p + "pwconv1.weight", (size_t)4 * C * C);
    b.pw1 = pack_conv(c, p + "pwconv1.weight", p + "pwconv1.bias", 4 * C, C);
    const float* gamma = host_param(c, p + "gamma", C);   // layer scale folded into pwconv2 (convnext.py:50-51)
    b.pw2 = pack_conv(c, p + "pwconv2.weight", p + "pwconv2.bias", C, 4 * C, 1, 1, gamma);
    // fused pwconv1 -> GELU -> pwconv2 -> + residual (mlp_fused.hip) for the narrow blocks of the f16x2 mode; UNI_NO_MLP_FUSED = A/B switch
    static const bool mlp_off = getenv("UNI_NO_MLP_FUSED") != nullptr;
    const float* w2 = host_param(c, p + "pwconv2.weight", (size_t)4 * C * C);
    if (c->b32 == FMT_H2 && mlp_fused_supported(C) && !mlp_off && w1 && w2 && b.pw1.bias && b.pw2.bias) {
        std::vector<uint16_t> blob(mlp_blob_bytes(C) / 2);
        // layout 1 (16-row waves, two per SIMD) where it exists; UNI_MLP_LAYOUT=0 forces the 32-row kernel (A/B switch)
        static const char* lay_env = getenv("UNI_MLP_LAYOUT");
        b.mlp_layout = (mlp_fused16_supported(C) && !(lay_env && lay_env[0] == '0')) ? 1 : 0;
        if (b.mlp_layout) mlp_pack16_host(w1, w2, gamma, C, blob.data(), &b.mlp_ws1, &b.mlp_ws2);
        else mlp_pack_host(w1, w2, gamma, C, blob.data(), &b.mlp_ws1, &b.mlp_ws2);
        b.mlp_blob = dev_upload<uint16_t>(c, blob.data(), blob.size());
    }
    return b;
}

static PBaseConv pack_baseconv(uni_ctx* c, const std::string& p, int cin, int cout, int k, int stride) {
    PBaseConv b;
    b.k = k; b.stride = stride;
    b.conv = pack_conv(c, p + "conv.weight", "", cout, cin, k, k);
    b.gn = up_affine(c, p + "bn.", cout);
    return b;
}

static PAffine concat_affine(uni_ctx* c, const std::vector<std::string>& prefixes, int C) {
    std::vector<float> g, b;
    for (auto& p : prefixes) {
        const float* hg = host_param(c, p + "weight", C);
        const float* hb = host_param(c, p + "bias", C);
        for (int i = 0; i < C; ++i) { g.push_back(hg ? hg[i] : 0.f); b.push_back(hb ? hb[i] : 0.f); }
    }
    PAffine a;
    a.g = dev_upload<float>(c, g.data(), g.size());
    a.b = dev_upload<float>(c, b.data(), b.size());
    return a;
}

static PCsp pack_csp(uni_ctx* c, const std::string& p, int cin, int cout) {
    PCsp s;
    s.cin = cin; s.cout = cout; s.h = cout / 2;
    // conv1 | conv2 share their input -> one GEMM with N = 2h (network_blocks.py:180-182)
    s.c12 = pack_convs(c, {ConvSrc{p + "conv1.conv.weight", "", s.h}, ConvSrc{p + "conv2.conv.weight", "", s.h}}, cin, 1, 1);
    s.gn12 = concat_affine(c, {p + "conv1.bn.", p + "conv2.bn."}, s.h);
    for (int i = 0; i < 3; ++i) {
        s.m1[i] = pack_baseconv(c, p + "m." + std::to_string(i) + ".conv1.", s.h, s.h, 1, 1);
        s.m2[i] = pack_baseconv(c, p + "m." + std::to_string(i) + ".conv2.", s.h, s.h, 3, 1);
    }
    s.c3 = pack_baseconv(c, p + "conv3.", 2 * s.h, cout, 1, 1);
    return s;
}

int engine_finalize(uni_ctx* c) {
    UNI_CHECK_HIP(hipSetDevice(c->device));
    const auto& cfg = c->cfg;
    const int* d = cfg.dims;
    const std::string bb = "backbone.backbone.";
    // ---- ConvNeXt ----
    {
        const float* w = host_param(c, bb + "downsample_layers.0.0.weight", (size_t)d[0] * 48);
        std::vector<float> t((size_t)48 * d[0], 0.f);
        if (w)
            for (int n = 0; n < d[0]; ++n)
                for (int k = 0; k < 48; ++k) t[(size_t)k * d[0] + n] = w[(size_t)n * 48 + k];
        c->stem_w = dev_upload<float>(c, t.data(), t.size());
        c->stem_b = up_vec(c, bb + "downsample_layers.0.0.bias", d[0]);
        c->stem_ln = up_affine(c, bb + "downsample_layers.0.1.", d[0]);
    }
    for (int i = 1; i < 4; ++i) {
        std::string p = bb + "downsample_layers." + std::to_string(i) + ".";
        c->ds_ln[i] = up_affine(c, p + "0.", d[i - 1]);
        c->ds_conv[i] = pack_conv(c, p + "1.weight", p + "1.bias", d[i], d[i - 1], 2, 2);
    }
    for (int i = 0; i < 4; ++i) {
        c->blocks[i].clear();
        for (int j = 0; j < cfg.depths[i]; ++j)
            c->blocks[i].push_back(pack_block(c, bb + "stages." + std::to_string(i) + "." + std::to_string(j) + ".", d[i]));
    }
    for (int i = 1; i < 4; ++i) c->out_norm[i] = up_affine(c, bb + "norm" + std::to_string(i) + ".", d[i]);
    // ---- PAFPN ----
    const int c0 = d[1], c1 = d[2], c2 = d[3];
    c->lateral0 = pack_baseconv(c, "backbone.lateral_conv0.", c2, c1, 1, 1);
    c->c3p4 = pack_csp(c, "backbone.C3_p4.", 2 * c1, c1);
    c->reduce1 = pack_baseconv(c, "backbone.reduce_conv1.", c1, c0, 1, 1);
    c->c3p3 = pack_csp(c, "backbone.C3_p3.", 2 * c0, c0);
    c->bu2 = pack_baseconv(c, "backbone.bu_conv2.", c0, c0, 3, 2);
    c->c3n3 = pack_csp(c, "backbone.C3_n3.", 2 * c0, c1);
    c->bu1 = pack_baseconv(c, "backbone.bu_conv1.", c1, c1, 3, 2);
    c->c3n4 = pack_csp(c, "backbone.C3_n4.", 2 * c1, c2);
    // ---- head ----
    const int ch[3] = {c0, c1, c2};
    for (int k = 0; k < 3; ++k) {
        const std::string ks = std::to_string(k);
        c->stems[k] = pack_baseconv(c, "head.stems." + ks + ".", ch[k], 256, 1, 1);
        c->beta[k] = up_vec(c, "head.beta_" + ks, 256);
        c->att[k].clear();
        for (int n = 0; n < cfg.n_layer_att; ++n)
            c->att[k].push_back(pack_block(c, "head.att_layers." + ks + "." + std::to_string(n) + ".", 256));
        c->tower0[k] = pack_convs(c, {ConvSrc{"head.cls_convs." + ks + ".0.conv.weight", "", 256},
                                      ConvSrc{"head.reg_convs." + ks + ".0.conv.weight", "", 256}}, 256, 3, 3);
        c->tower0_gn[k] = concat_affine(c, {"head.cls_convs." + ks + ".0.bn.", "head.reg_convs." + ks + ".0.bn."}, 256);
        for (int i = 1; i < 4; ++i) {
            c->cls_convs[k][i] = pack_baseconv(c, "head.cls_convs." + ks + "." + std::to_string(i) + ".", 256, 256, 3, 1);
            c->reg_convs[k][i] = pack_baseconv(c, "head.reg_convs." + ks + "." + std::to_string(i) + ".", 256, 256, 3, 1);
        }
        c->cls_pred[k] = pack_conv(c, "head.cls_preds." + ks + ".weight", "head.cls_preds." + ks + ".bias", cfg.num_classes, 256);
        c->cls_pred_sot[k] = pack_conv(c, "head.cls_preds_sot." + ks + ".weight", "head.cls_preds_sot." + ks + ".bias", 1, 256);
        c->regobj[k] = pack_convs(c, {ConvSrc{"head.reg_preds." + ks + ".weight", "head.reg_preds." + ks + ".bias", 4},
                                      ConvSrc{"head.obj_preds." + ks + ".weight", "head.obj_preds." + ks + ".bias", 1}}, 256, 1, 1);
        c->regobj_sot[k] = pack_convs(c, {ConvSrc{"head.reg_preds_sot." + ks + ".weight", "head.reg_preds_sot." + ks + ".bias", 4},
                                          ConvSrc{"head.obj_preds_sot." + ks + ".weight", "head.obj_preds_sot." + ks + ".bias", 1}}, 256, 1, 1);
        if (cfg.mask)
            c->controllers[k] = pack_conv(c, "head.controllers." + ks + ".weight", "head.controllers." + ks + ".bias", 169, 256, 3, 3);
    }
    if (cfg.mask) {
        const std::string mb = "head.mask_branch.";
        for (int k = 0; k < 3; ++k) {
            c->refine[k] = pack_conv(c, mb + "refine." + std::to_string(k) + ".0.weight", "", 128, ch[k], 3, 3);
            c->refine_gn[k] = up_affine(c, mb + "refine." + std::to_string(k) + ".1.", 128);
        }
        for (int i = 0; i < 4; ++i) {
            c->mtower[i] = pack_conv(c, mb + "tower." + std::to_string(i) + ".0.weight", "", 128, 128, 3, 3);
            c->mtower_gn[i] = up_affine(c, mb + "tower." + std::to_string(i) + ".1.", 128);
        }
        c->mtower_out = pack_conv(c, mb + "tower.4.weight", mb + "tower.4.bias", 8, 128);
        c->upm0 = pack_conv(c, mb + "up_mask_layer.0.weight", mb + "up_mask_layer.0.bias", 128, 128, 3, 3);
        c->upm1 = pack_conv(c, mb + "up_mask_layer.2.weight", mb + "up_mask_layer.2.bias", 9 * cfg.up_rate * cfg.up_rate, 128);
    }
    // ---- interaction / embedding ----
    c->bott = pack_conv(c, "bottleneck.0.weight", "bottleneck.0.bias", 256, c1);
    c->bott_gn = up_affine(c, "bottleneck.1.", 256);
    const std::string e = "transformer.encoder.layers.0.";
    c->value_proj = pack_conv(c, e + "self_attn.value_proj.weight", e + "self_attn.value_proj.bias", 256, 256);
    c->offaw = pack_convs(c, {ConvSrc{e + "self_attn.sampling_offsets.weight", e + "self_attn.sampling_offsets.bias", 128},
                              ConvSrc{e + "self_attn.attention_weights.weight", e + "self_attn.attention_weights.bias", 64}}, 256, 1, 1);
    c->output_proj = pack_conv(c, e + "self_attn.output_proj.weight", e + "self_attn.output_proj.bias", 256, 256);
    c->lin1 = pack_conv(c, e + "linear1.weight", e + "linear1.bias", 1024, 256);
    c->lin2 = pack_conv(c, e + "linear2.weight", e + "linear2.bias", 256, 1024);
    c->norm1 = up_affine(c, e + "norm1.", 256);
    c->norm2 = up_affine(c, e + "norm2.", 256);
    c->level_embed = up_vec(c, "transformer.level_embed", 512);
    c->up1 = pack_conv(c, "upsample_layer.1.weight", "upsample_layer.1.bias", 256, 64, 3, 3);
    c->up3 = pack_conv(c, "upsample_layer.3.weight", "upsample_layer.3.bias", cfg.embed_dim, 256, 3, 3);
    c->pos_row = up_vec(c, "pos_emb.row_embed.weight", 40 * 128);
    c->pos_col = up_vec(c, "pos_emb.col_embed.weight", 40 * 128);
    // GroupNorm statistics arena (zeroed at the start of every stage call)
    UNI_CHECK_HIP(hipMalloc(&c->stats, UNI_STATS_SLOTS * 64 * sizeof(double)));
    c->dev_allocs.push_back(c->stats);
    for (int i = 0; i < 3; ++i) {
        UNI_CHECK_HIP(hipStreamCreateWithFlags(&c->aux[i], hipStreamNonBlocking));
        UNI_CHECK_HIP(hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming));
    }
    UNI_CHECK_HIP(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    UNI_REQUIRE(!c->failed, "finalize: device upload failed");
    c->host.clear();      // host copies are no longer needed
    c->finalized = true;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------------------
int engine_reserve(uni_ctx* c, int B, int H, int W) {
    const size_t need = (size_t)B * H * W * 3200 + ((size_t)96 << 20) + 4 * (UNI_SLAB_BYTES + 4096);   // >= head: 3 level slices (H*W/64*14336 B each) + casts + mask branch + split-K slabs
    if (need <= c->ws_cap) return 0;
    UNI_CHECK_HIP(hipSetDevice(c->device));
    UNI_CHECK_HIP(hipDeviceSynchronize());    // growing is rare; never happens inside a timed loop after warm-up
    if (c->ws) UNI_CHECK_HIP(hipFree(c->ws));
    c->ws = nullptr;
    c->ws_cap = 0;
    UNI_CHECK_HIP(hipMalloc(&c->ws, need));
    c->ws_cap = need;
    return 0;
}

template <class T>
static T* wsalloc(uni_ctx* c, size_t n) {
    size_t off = (c->ws_off + 255) & ~(size_t)255;
    size_t bytes = n * sizeof(T);
    if (off + bytes > c->ws_cap) {
        c->ws_overflow = true;
        uni_set_error("workspace overflow: need %zu more bytes (cap %zu)", off + bytes - c->ws_cap, c->ws_cap);
        return reinterpret_cast<T*>(c->ws);   // keep pointers valid; the stage returns an error before launching more
    }
    c->ws_off = off + bytes;
    return reinterpret_cast<T*>(c->ws + off);
}

static int stage_begin(uni_ctx* c, int B, int H, int W, hipStream_t s) {
    UNI_REQUIRE(c && c->finalized, "context not finalized");
    UNI_CHECK_HIP(hipSetDevice(c->device));
    int rc = engine_reserve(c, B, H, W);
    if (rc) return rc;
    c->nb = B;
    c->ws_off = 0;
    c->ws_overflow = false;
    c->stats_slot = 0;
    UNI_CHECK_HIP(hipMemsetAsync(c->stats, 0, UNI_STATS_SLOTS * 64 * sizeof(double), s));
    return 0;
}
static double* next_stats(uni_ctx* c, int n = 1) {   // n consecutive slots (one per sample)
    if (c->stats_slot + n > UNI_STATS_SLOTS) { c->ws_overflow = true; uni_set_error("GN stats arena exhausted"); return c->stats; }
    double* p = c->stats + (size_t)c->stats_slot * 64;
    c->stats_slot += n;
    return p;
}
#define RUN(expr) do { int _rc = (expr); if (_rc) return _rc; if (c->ws_overflow) return -3; } while (0)


// ------------------------------------------------------------------------------------------------
// optional per-kernel-class timing with HIP events on the launch stream (bench.py roofline leg)
// ------------------------------------------------------------------------------------------------
enum { PC_GEMM = 0, PC_DWLN = 1, PC_GN = 2, PC_LN = 3, PC_MISC = 4, PC_NCLS = 5 };
template <class F>
static int prof_run(uni_ctx* c, int cls, double work, hipStream_t s, F&& f) {
    if (!c->prof_on) return f();
    ProfRec r;
    r.cls = cls; r.work = work;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return f();
    (void)hipEventRecord(r.a, s);
    int rc = f();
    (void)hipEventRecord(r.b, s);
    c->recs.push_back(r);
    return rc;
}
// ------------------------------------------------------------------------------------------------
// operand-range check (uni_ctx_set_check / UNI_CHECK_SAT=1): the f16x2 operand format saturates at +-65504 (common.h h2_split).  Trained
// checkpoints cannot be validated offline, so a context in check mode scans every operand buffer it produces and counts the saturated
// elements; uni_ctx_stats reads the counters.  Off by default (one extra pass over every operand buffer).
// ------------------------------------------------------------------------------------------------
__global__ void sat_scan_kernel(const unsigned short* buf, long rows, int C, int ld, unsigned long long* cnt) {
    const int groups = C >> 3;
    const long total = rows * groups;
    unsigned n = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / groups;
        const int g = (int)(i - r * groups);
        const unsigned short* p = buf + (r * ld + 8 * g) * 2;          // [8 x hi][8 x lo] per 8 channels
#pragma unroll
        for (int e = 0; e < 8; ++e) n += (p[e] & 0x7fffu) >= 0x7bffu ? 1u : 0u;   // |hi| == 65504 (or non-finite)
    }
    n = (unsigned)wave_sum((float)n);      // < 2^24 per wave: exact
    if ((threadIdx.x & 63) == 0 && n) atomicAdd(&cnt[0], (unsigned long long)n);
    if (blockIdx.x == 0 && threadIdx.x == 0) { atomicAdd(&cnt[1], (unsigned long long)(rows * (long)C)); atomicAdd(&cnt[2], 1ull); }
}
static void sat_scan(uni_ctx* c, const void* buf, long rows, int C, int ld, hipStream_t s) {
    if (!c->check_sat || !c->sat_dev || c->b32 != FMT_H2 || !buf || rows <= 0 || C < 8) return;
    const long total = rows * (C >> 3);
    int grid = (int)std::min<long>((total + 255) / 256, 2048);
    hipLaunchKernelGGL(sat_scan_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<const unsigned short*>(buf), rows, C, ld, c->sat_dev);
}
int engine_set_check(uni_ctx* c, int on) {
    UNI_REQUIRE(c, "ctx is NULL");
    UNI_CHECK_HIP(hipSetDevice(c->device));
    if (on && !c->sat_dev) {
        UNI_CHECK_HIP(hipMalloc(&c->sat_dev, 4 * sizeof(unsigned long long)));
        c->dev_allocs.push_back(c->sat_dev);
    }
    if (c->sat_dev) UNI_CHECK_HIP(hipMemset(c->sat_dev, 0, 4 * sizeof(unsigned long long)));
    c->check_sat = on != 0;
    return 0;
}
int engine_stats(uni_ctx* c, long long* out4) {
    UNI_REQUIRE(c && out4, "stats: NULL argument");
    for (int i = 0; i < 4; ++i) out4[i] = 0;
    if (!c->sat_dev) return 0;
    UNI_CHECK_HIP(hipSetDevice(c->device));
    UNI_CHECK_HIP(hipDeviceSynchronize());
    unsigned long long h[4];
    UNI_CHECK_HIP(hipMemcpy(h, c->sat_dev, sizeof(h), hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; ++i) out4[i] = (long long)h[i];
    return 0;
}

static int p_gemm(uni_ctx* c, const GemmArgs& g, hipStream_t s) {
    if (c->prof_on) {   // algorithmic bytes: input map + weights + every output/residual stream, each once
        const double es = act_elem_bytes(g.b32);
        c->prof_bytes += (double)g.Hin * g.Win * g.Cin * es + (double)g.N * g.K * es +
                         (double)g.M * g.N * ((g.outF ? 4.0 : 0.0) + (g.outB ? es : 0.0) + (g.res ? 4.0 : 0.0));
    }
    const size_t before = c->recs.size();
    int rc = prof_run(c, PC_GEMM, 2.0 * g.M * g.N * g.K, s, [&] { return launch_gemm(g, s); });
    if (c->recs.size() > before) { ProfRec& r = c->recs.back(); r.M = g.M; r.N = g.N; r.K = g.K; r.conv = g.KH * 10 + g.stride; }
    if (g.outB) sat_scan(c, g.outB, g.M, g.N, g.ldb, s);
    return rc;
}
static int p_dwln(uni_ctx* c, DwLnArgs d, hipStream_t s) {
    d.b32 = c->b32;
    int rc = prof_run(c, PC_DWLN, (double)(d.B > 0 ? d.B : 1) * d.H * d.W * d.C * (4.0 + act_elem_bytes(c->b32)) + 49.0 * d.C * 4, s, [&] { return launch_dwconv7_ln(d, s); });
    sat_scan(c, d.out, (long)(d.B > 0 ? d.B : 1) * d.H * d.W, d.C, d.C, s);
    return rc;
}
static int p_gn(uni_ctx* c, GnApplyArgs a, hipStream_t s) {
    a.b32 = c->b32;
    double b = (double)(a.B > 0 ? a.B : 1) * a.M * a.C * (4.0 + (a.outF ? 4 : 0) + (a.outB ? 2 : 0) + (a.outUp ? 8 : 0));
    int rc = prof_run(c, PC_GN, b, s, [&] { return launch_gn_apply(a, s); });
    if (a.outB) sat_scan(c, a.outB, (long)(a.B > 0 ? a.B : 1) * a.M, a.C, a.ldb, s);
    return rc;
}
static int p_ln(uni_ctx* c, LnArgs a, hipStream_t s) {
    a.b32 = c->b32;
    double b = (double)a.M * a.C * (4.0 + (a.outF ? 4 : 0) + (a.outB ? 2 : 0));
    int rc = prof_run(c, PC_LN, b, s, [&] { return launch_layernorm(a, s); });
    if (a.outB && !a.ps_h) sat_scan(c, a.outB, a.M, a.C, a.ldb, s);
    return rc;
}
int engine_prof_begin(uni_ctx* c) {
    for (auto& r : c->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    c->recs.clear();
    c->prof_on = true;
    c->prof_bytes = 0.0;
    return 0;
}
int engine_prof_end(uni_ctx* c, double* out) {   // out[PC_NCLS][3] = {ms, work, launches}; out[15] = GEMM algorithmic bytes
    c->prof_on = false;
    UNI_CHECK_HIP(hipDeviceSynchronize());
    for (int i = 0; i < PC_NCLS * 3; ++i) out[i] = 0.0;
    out[PC_NCLS * 3] = c->prof_bytes;
    // UNI_PROF_DUMP=<path>: one line per profiled launch (class, M, N, K, conv code KH*10+stride, us, work)
    FILE* dump = getenv("UNI_PROF_DUMP") ? fopen(getenv("UNI_PROF_DUMP"), "a") : nullptr;
    for (auto& r : c->recs) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        if (dump) fprintf(dump, "%d %d %d %d %d %.2f %.6g\n", r.cls, r.M, r.N, r.K, r.conv, ms * 1e3, r.work);
        out[r.cls * 3 + 0] += ms; out[r.cls * 3 + 1] += r.work; out[r.cls * 3 + 2] += 1;
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    if (dump) fclose(dump);
    c->recs.clear();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// building blocks
// ------------------------------------------------------------------------------------------------
// activation (GEMM-operand) buffer pointer: bf16 elements by default, fp32 in the exact-fp32 precision mode
struct ActPtr {
    char* p = nullptr; int es = 2;
    ActPtr() {}
    ActPtr(char* p_, int es_) : p(p_), es(es_) {}
    ActPtr operator+(size_t elems) const { return ActPtr(p + elems * es, es); }
    operator bf16*() const { return reinterpret_cast<bf16*>(p); }
    bool operator==(const ActPtr& o) const { return p == o.p; }
};
static ActPtr actalloc(uni_ctx* c, size_t n) { const int es = act_elem_bytes(c->b32); return ActPtr(wsalloc<char>(c, n * es), es); }

struct Out {
    float* F = nullptr; int ldf = 0;
    ActPtr B; int ldb = 0;
    ActPtr Up; int ldu = 0;
    const float* prior = nullptr; const float* pbeta = nullptr;
};

static GemmArgs conv_args(const PConv& p, ActPtr A, int lda, int Hin, int Win, int stride, int pad, int B = 1) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = p.W;
    const int Hout = (Hin + 2 * pad - p.KH) / stride + 1, Wout = (Win + 2 * pad - p.KW) / stride + 1;
    g.Mper = Hout * Wout; g.M = B * g.Mper; g.N = p.N; g.K = p.K; g.Kpad = p.Kpad;
    g.Hin = Hin; g.Win = Win; g.Cin = p.Cin; g.KH = p.KH; g.KW = p.KW; g.stride = stride; g.pad = pad; g.Wout = Wout;
    g.bias = p.bias;
    g.b32 = p.b32; g.wscale = p.wscale;
    return g;
}

// Single-frame (small M) problems leave most of the 256 CUs without a tile: K ranges of one tile then go to different blocks, every
// block stores its partial tile into a slab and `splitk_reduce_kernel` sums the ranges in a fixed order (deterministic; bias / residual /
// GroupNorm sums are applied there -- gemm_h2.hip, GemmArgs::splitk).  Returns the number of K ranges (1 = no split).
// UNI_NO_SPLITK = A/B switch.
static int choose_splitk(const uni_ctx* c, const GemmArgs& g) {
    static const bool off = getenv("UNI_NO_SPLITK") != nullptr;
    if (off || c->b32 != FMT_H2 || g.N < 128 || g.N % 4 != 0 || g.K % 32 != 0) return 1;
    const bool conv = g.KH != 1 || g.KW != 1 || g.stride != 1 || g.pad != 0;
    const long tiles = (long)cdiv(g.M, 128) * cdiv(g.N, 128);
    const int nk = g.K / 32;
    // tools/gemm_b1_bench.py on the single-frame shapes: 3x3 convolutions gain 25-50 % from ~400 blocks of >= 12 K steps (4000 x 384 x 3456:
    // 79 -> 60 us with 4 ranges, 1000 x 768 x 6912: 140 -> 69 us with 8); plain GEMMs only with very long K and few tiles (1000 x 1536 x 6144:
    // 107 -> 94 us); 1x1 convolutions and the stage-2 MLP (N >= 768, hundreds of 64 x 64 tiles) do not
    int sk = 1;
    if (gemm_h2d_choice(g)) {
        // deep-pipeline 64 x 64 tiles (3 blocks per CU): only problems with < ~200 of them still leave CUs empty (1000 x 256 x 2304: 26 -> 22 us,
        // 1000 x 768 x 6912: 73 -> 58 us with 3 ranges); everything larger is faster unsplit (no slab round trip)
        const long t64 = (long)cdiv(g.M, 64) * cdiv(g.N, 64);
        if (t64 > 200 || nk < 36) return 1;
        sk = 3;
    } else if (conv) {
        if (tiles > 128 || nk < 24) return 1;
        sk = std::min((int)std::min<long>((400 + tiles / 2) / tiles, nk / 12), 8);
    } else if (tiles <= 96 && nk >= 192) sk = 4;      // only reached with UNI_NO_H2D (gemm_h2d_choice claims every problem this small): plain long-K GEMMs on the generic tiles
    while (sk > 1 && (size_t)sk * g.M * g.N * sizeof(float) > UNI_SLAB_BYTES) --sk;      // the partial-tile slab has a fixed budget in the workspace plan
    return sk >= 2 ? sk : 1;
}

// conv (no act) -> GroupNorm(G) -> act, written to `o`
static int run_conv_gn(uni_ctx* c, const PConv& conv, const PAffine& gn, int G, float eps, int act, ActPtr A, int lda,
                       int Hin, int Win, int stride, const Out& o, hipStream_t s) {
    const size_t mark = c->ws_off;
    const int B = c->nb;
    GemmArgs g = conv_args(conv, A, lda, Hin, Win, stride, (conv.KH - 1) / 2, B);
    float* raw = wsalloc<float>(c, (size_t)g.M * g.N);
    double* st = next_stats(c, B);
    g.outF = raw; g.ldf = g.N;
    g.stats = st; g.cpg = g.N / G;
    const int sk = choose_splitk(c, g);
    if (sk > 1) { g.splitk = sk; g.slab = wsalloc<float>(c, (size_t)sk * g.M * g.N); }     // K ranges -> slab -> reduce (+ bias, group sums)
    RUN(p_gemm(c, g, s));
    GnApplyArgs a;
    a.x = raw; a.ldx = g.N; a.stats = st; a.gamma = gn.g; a.beta = gn.b; a.eps = eps;
    a.M = g.Mper; a.B = B; a.C = g.N; a.G = G; a.act = act;
    a.prior = o.prior; a.prior_beta = o.pbeta;
    a.outF = o.F; a.ldf = o.ldf; a.outB = o.B; a.ldb = o.ldb; a.outUp = o.Up; a.ldu = o.ldu;
    a.W = (Win + 2 * ((conv.KW - 1) / 2) - conv.KW) / stride + 1;
    RUN(p_gn(c, a, s));
    c->ws_off = mark;     // raw is dead once gn_apply is enqueued (single in-order stream)
    return 0;
}
static int run_baseconv(uni_ctx* c, const PBaseConv& b, ActPtr A, int lda, int Hin, int Win, const Out& o, hipStream_t s) {
    return run_conv_gn(c, b.conv, b.gn, 16, 1e-3f, ACT_SILU, A, lda, Hin, Win, b.stride, o, s);
}

// ConvNeXt block on the fp32 residual stream x [H*W][C] (in place); t/hid are caller-provided scratch
static int run_block(uni_ctx* c, const PBlock& b, float* x, int H, int W, ActPtr t, ActPtr hid, ActPtr outB, hipStream_t s) {
    const int C = b.C, M = H * W * c->nb;
    const size_t mark = c->ws_off;
    {
        DwLnArgs d;
        d.x = x; d.w = b.dw_w; d.bias = b.dw_b; d.gamma = b.ln.g; d.beta = b.ln.b; d.eps = 1e-6f;
        d.H = H; d.W = W; d.C = C; d.B = c->nb; d.out = t;
        const size_t before = c->recs.size();
        RUN(p_dwln(c, d, s));
        if (c->recs.size() > before) { ProfRec& r = c->recs.back(); r.M = M; r.N = C; r.K = W; }
    }
    // one launch, hidden activations kept in registers; below ~192 row tiles of 128 (most CUs idle) the two GEMMs with their 64 / 128-row
    // tiles fill the chip better (measured at M = 16000: 96 vs 89 us, at M = 1000: 86 vs ~50 us)
    if (b.mlp_blob && M >= 192 * 128 && !c->check_sat) {   // (check mode takes the two-launch path: the hidden tensor must exist to be scanned)
        MlpArgs m;
        m.A = t.p; m.lda = C; m.blob = b.mlp_blob; m.b1 = b.pw1.bias; m.b2 = b.pw2.bias; m.ws1 = b.mlp_ws1; m.ws2 = b.mlp_ws2;
        m.res = x; m.ldr = C; m.out = x; m.ldo = C; m.outB = outB.p; m.ldb = C; m.M = M; m.C = C; m.layout = b.mlp_layout;
        if (c->prof_on) c->prof_bytes += (double)M * C * (4.0 + 8.0 + (outB.p ? 4.0 : 0.0)) + (double)mlp_blob_bytes(C);
        const size_t before = c->recs.size();
        RUN(prof_run(c, PC_GEMM, 2.0 * 2.0 * M * 4.0 * C * C, s, [&] { return launch_mlp_fused(m, s); }));
        if (c->recs.size() > before) { ProfRec& r = c->recs.back(); r.M = M; r.N = C; r.K = 4 * C; r.conv = 99; }
        c->ws_off = mark;
        return 0;
    }
    GemmArgs g1 = conv_args(b.pw1, t, C, M, 1, 1, 0);
    g1.act = ACT_GELU; g1.outB = hid; g1.ldb = 4 * C;
    RUN(p_gemm(c, g1, s));
    GemmArgs g2 = conv_args(b.pw2, hid, 4 * C, M, 1, 1, 0);
    g2.res = x; g2.ldr = C; g2.outF = x; g2.ldf = C; g2.outB = outB; g2.ldb = C;
    const int sk2 = outB.p ? 1 : choose_splitk(c, g2);
    if (sk2 > 1) { g2.splitk = sk2; g2.slab = wsalloc<float>(c, (size_t)sk2 * M * C); }
    RUN(p_gemm(c, g2, s));
    c->ws_off = mark;
    return 0;
}

// CSP layer: cat buffer `in` [M][cin] bf16 -> `o`
static int run_csp(uni_ctx* c, const PCsp& p, ActPtr in, int H, int W, const Out& o, hipStream_t s) {
    const int M = H * W * c->nb, h = p.h;
    const size_t mark = c->ws_off;
    ActPtr cat = actalloc(c, (size_t)M * 2 * h);     // [x_1 | x_2]
    ActPtr t1 = actalloc(c, (size_t)M * h);
    ActPtr t2 = actalloc(c, (size_t)M * h);
    Out o12; o12.B = cat; o12.ldb = 2 * h;
    RUN(run_conv_gn(c, p.c12, p.gn12, 32, 1e-3f, ACT_SILU, in, p.cin, H, W, 1, o12, s));
    ActPtr cur = cat; int ld = 2 * h;
    for (int i = 0; i < 3; ++i) {
        Out a; a.B = t1; a.ldb = h;
        RUN(run_baseconv(c, p.m1[i], cur, ld, H, W, a, s));
        Out b2;
        if (i == 2) { b2.B = cat; b2.ldb = 2 * h; } else { b2.B = t2; b2.ldb = h; }
        RUN(run_baseconv(c, p.m2[i], t1, h, H, W, b2, s));
        cur = t2; ld = h;
    }
    RUN(run_baseconv(c, p.c3, cat, 2 * h, H, W, o, s));
    c->ws_off = mark;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// stage: backbone + PAFPN   (unicorn.py:231-258)
// ------------------------------------------------------------------------------------------------
int engine_backbone_fpn(uni_ctx* c, const float* img, int B, int H, int W, float* fpn0, float* fpn1, float* fpn2, float* feat16,
                        hipStream_t s) {
    UNI_REQUIRE(H % 32 == 0 && W % 32 == 0 && H > 0 && W > 0, "backbone: H=%d W=%d must be positive multiples of 32", H, W);
    UNI_REQUIRE(B >= 1 && B <= 64, "backbone: batch %d unsupported (1..64)", B);
    RUN(stage_begin(c, B, H, W, s));
    const int* d = c->cfg.dims;
    const int c0 = d[1], c1 = d[2], c2 = d[3];
    const int H8 = H / 8, W8 = W / 8, H16 = H / 16, W16 = W / 16, H32 = H / 32, W32 = W / 32;
    const size_t M8 = (size_t)B * H8 * W8, M16 = (size_t)B * H16 * W16, M32 = (size_t)B * H32 * W32;   // rows over the batch
    // persistent (for this call) PAFPN inputs
    ActPtr cat8 = actalloc(c, M8 * 2 * c0);      // [up(fpn_out1) | x2]
    ActPtr cat16a = actalloc(c, M16 * 2 * c1);   // [up(fpn_out0) | x1]
    ActPtr cat16b = actalloc(c, M16 * 2 * c0);   // [p_out1 | fpn_out1]
    ActPtr cat32 = actalloc(c, M32 * 2 * c1);    // [p_out0 | fpn_out0]
    ActPtr x0b = actalloc(c, M32 * c2);
    // ---- ConvNeXt ----
    {
        int Hs = H / 4, Ws = W / 4;
        float* x = wsalloc<float>(c, (size_t)B * Hs * Ws * d[0]);
        StemArgs st;
        st.img = img; st.H = H; st.W = W; st.B = B; st.w = c->stem_w; st.bias = c->stem_b; st.gamma = c->stem_ln.g; st.beta = c->stem_ln.b;
        st.C = d[0]; st.out = x;
        RUN(prof_run(c, PC_MISC, 0.0, s, [&] { return launch_stem(st, s); }));
        for (int i = 0; i < 4; ++i) {
            if (i > 0) {
                // LN_cf + conv2x2/s2 (convnext.py:80-86)
                ActPtr t = actalloc(c, (size_t)B * Hs * Ws * d[i - 1]);
                LnArgs ln;
                ln.x = x; ln.ldx = d[i - 1]; ln.gamma = c->ds_ln[i].g; ln.beta = c->ds_ln[i].b; ln.eps = 1e-6f;
                ln.M = B * Hs * Ws; ln.C = d[i - 1]; ln.outB = t; ln.ldb = d[i - 1];
                RUN(p_ln(c, ln, s));
                float* xn = wsalloc<float>(c, (size_t)B * (Hs / 2) * (Ws / 2) * d[i]);
                GemmArgs g = conv_args(c->ds_conv[i], t, d[i - 1], Hs, Ws, 2, 0, B);
                g.outF = xn; g.ldf = d[i];
                RUN(p_gemm(c, g, s));
                x = xn; Hs /= 2; Ws /= 2;
            }
            const size_t M = (size_t)B * Hs * Ws;
            const int C = d[i];
            ActPtr t = actalloc(c, M * C);
            ActPtr hid = actalloc(c, M * 4 * C);
            for (auto& b : c->blocks[i]) RUN(run_block(c, b, x, Hs, Ws, t, hid, ActPtr(), s));
            if (i >= 1) {
                LnArgs ln;
                ln.x = x; ln.ldx = C; ln.gamma = c->out_norm[i].g; ln.beta = c->out_norm[i].b; ln.eps = 1e-6f;
                ln.M = (int)M; ln.C = C;
                if (i == 1) { ln.outB = cat8 + c0; ln.ldb = 2 * c0; }
                else if (i == 2) { ln.outB = cat16a + c1; ln.ldb = 2 * c1; ln.outF = feat16; ln.ldf = c1; }
                else { ln.outB = x0b; ln.ldb = c2; }
                RUN(p_ln(c, ln, s));
            }
        }
    }
    // ---- PAFPN (yolo_pafpn_new.py:132-161) ----
    {
        Out o;  // fpn_out0 -> cat32[:, c1:] and 2x-upsampled into cat16a[:, :c1]
        o.B = cat32 + c1; o.ldb = 2 * c1; o.Up = cat16a; o.ldu = 2 * c1;
        RUN(run_baseconv(c, c->lateral0, x0b, c2, H32, W32, o, s));
    }
    ActPtr f_out0 = actalloc(c, M16 * c1);
    { Out o; o.B = f_out0; o.ldb = c1; RUN(run_csp(c, c->c3p4, cat16a, H16, W16, o, s)); }
    {
        Out o;  // fpn_out1 -> cat16b[:, c0:] and upsampled into cat8[:, :c0]
        o.B = cat16b + c0; o.ldb = 2 * c0; o.Up = cat8; o.ldu = 2 * c0;
        RUN(run_baseconv(c, c->reduce1, f_out0, c1, H16, W16, o, s));
    }
    ActPtr pan2b = actalloc(c, M8 * c0);
    { Out o; o.F = fpn0; o.ldf = c0; o.B = pan2b; o.ldb = c0; RUN(run_csp(c, c->c3p3, cat8, H8, W8, o, s)); }
    { Out o; o.B = cat16b; o.ldb = 2 * c0; RUN(run_baseconv(c, c->bu2, pan2b, c0, H8, W8, o, s)); }
    ActPtr pan1b = actalloc(c, M16 * c1);
    { Out o; o.F = fpn1; o.ldf = c1; o.B = pan1b; o.ldb = c1; RUN(run_csp(c, c->c3n3, cat16b, H16, W16, o, s)); }
    { Out o; o.B = cat32; o.ldb = 2 * c1; RUN(run_baseconv(c, c->bu1, pan1b, c1, H16, W16, o, s)); }
    { Out o; o.F = fpn2; o.ldf = c2; RUN(run_csp(c, c->c3n4, cat32, H32, W32, o, s)); }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// stage: deformable interaction   (unicorn.py:260-276, deformable_transformer.py:58-131)
// feat_* are [B][hw][C2]; internally the tokens are laid out [B][2 frames][hw] so every sample is one contiguous
// (ref | cur) sequence for the sampler.
// ------------------------------------------------------------------------------------------------
int engine_interaction(uni_ctx* c, const float* feat_ref, const float* pos_ref, const float* feat_cur, const float* pos_cur,
                       int B, int h, int w, float* out_ref, float* out_cur, hipStream_t s) {
    UNI_REQUIRE(B >= 1 && B <= 32, "interaction: batch %d unsupported (1..32)", B);
    RUN(stage_begin(c, B, h * 16, w * 16, s));
    const int hw = h * w, C2 = c->cfg.dims[2];
    const size_t L = (size_t)2 * hw * B;                    // tokens over the batch
    ActPtr fb = actalloc(c, L * C2);
    RUN(prof_run(c, PC_MISC, 0.0, s, [&] { return launch_cast_operand_pair(feat_ref, feat_cur, fb, hw, C2, B, s, c->b32); }));     // tokens [B][ref | cur][hw]
    float* src = wsalloc<float>(c, L * 256);
    ActPtr srcb = actalloc(c, L * 256);
    {   // bottleneck: 1x1 conv + bias -> GroupNorm(32, eps 1e-5) per frame = 2B "samples" of hw rows
        c->nb = 2 * B;
        Out o; o.F = src; o.ldf = 256; o.B = srcb; o.ldb = 256;
        int rc = run_conv_gn(c, c->bott, c->bott_gn, 32, 1e-5f, ACT_NONE, fb, C2, hw, 1, 1, o, s);
        c->nb = B;
        if (rc) return rc;
    }
    ActPtr qb = actalloc(c, L * 256);
    RUN(prof_run(c, PC_MISC, 0.0, s, [&] { return launch_add_pos_bf16(src, pos_ref, pos_cur, c->level_embed, qb, hw, 256, s, c->b32, B); }));
    float* value = wsalloc<float>(c, L * 256);
    { GemmArgs g = conv_args(c->value_proj, srcb, 256, (int)L, 1, 1, 0); g.outF = value; g.ldf = 256; RUN(p_gemm(c, g, s)); }
    float* offaw = wsalloc<float>(c, L * 192);
    { GemmArgs g = conv_args(c->offaw, qb, 256, (int)L, 1, 1, 0); g.outF = offaw; g.ldf = 192; RUN(p_gemm(c, g, s)); }
    ActPtr attn = actalloc(c, L * 256);
    { MsdaFusedArgs m; m.value = value; m.offaw = offaw; m.ldo = 192; m.h = h; m.w = w; m.out = attn; m.b32 = c->b32; m.B = B; RUN(prof_run(c, PC_MISC, 0.0, s, [&] { return launch_msda_fused(m, s); })); }
    float* y = wsalloc<float>(c, L * 256);
    { GemmArgs g = conv_args(c->output_proj, attn, 256, (int)L, 1, 1, 0); g.res = src; g.ldr = 256; g.outF = y; g.ldf = 256; RUN(p_gemm(c, g, s)); }
    {   // src = norm1(src + attn)
        LnArgs ln; ln.x = y; ln.ldx = 256; ln.gamma = c->norm1.g; ln.beta = c->norm1.b; ln.eps = 1e-5f; ln.M = (int)L; ln.C = 256;
        ln.outF = src; ln.ldf = 256; ln.outB = srcb; ln.ldb = 256;
        RUN(p_ln(c, ln, s));
    }
    ActPtr hid = actalloc(c, L * 1024);
    { GemmArgs g = conv_args(c->lin1, srcb, 256, (int)L, 1, 1, 0); g.act = ACT_RELU; g.outB = hid; g.ldb = 1024; RUN(p_gemm(c, g, s)); }
    { GemmArgs g = conv_args(c->lin2, hid, 1024, (int)L, 1, 1, 0); g.res = src; g.ldr = 256; g.outF = y; g.ldf = 256; RUN(p_gemm(c, g, s)); }
    {   // src = norm2(src + ffn): tokens [B][2][hw] back to the two frame maps, one launch
        LnArgs ln; ln.x = y; ln.ldx = 256; ln.gamma = c->norm2.g; ln.beta = c->norm2.b; ln.eps = 1e-5f;
        ln.M = (int)L; ln.C = 256; ln.outF = out_ref; ln.outF2 = out_cur; ln.pair_hw = hw; ln.ldf = 256;
        RUN(p_ln(c, ln, s));
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// stage: embedding head   (unicorn.py:41-44,311-313)
// ------------------------------------------------------------------------------------------------
int engine_upsample(uni_ctx* c, const float* feat, int B, int h, int w, float* embed, hipStream_t s) {
    UNI_REQUIRE(B >= 1 && B <= 64, "upsample: batch %d unsupported (1..64)", B);
    RUN(stage_begin(c, B, h * 16, w * 16, s));
    const int H = 2 * h, W = 2 * w;
    const size_t M = (size_t)B * H * W;
    ActPtr ps = actalloc(c, M * 64);
    RUN(prof_run(c, PC_MISC, 0.0, s, [&] { return launch_pixel_shuffle_bf16(feat, ps, h, w, 256, s, c->b32, B); }));
    ActPtr mid = actalloc(c, M * 256);
    { GemmArgs g = conv_args(c->up1, ps, 64, H, W, 1, 1, B); g.act = ACT_RELU; g.outB = mid; g.ldb = 256; RUN(p_gemm(c, g, s)); }
    { GemmArgs g = conv_args(c->up3, mid, 256, H, W, 1, 1, B); g.outF = embed; g.ldf = c->cfg.embed_dim; RUN(p_gemm(c, g, s)); }
    return 0;
}

int engine_pos_embed(uni_ctx* c, int h, int w, float* out, hipStream_t s) {
    UNI_REQUIRE(c && c->finalized, "context not finalized");
    return launch_pos_embed(c->pos_row, c->pos_col, 40, 128, out, h, w, s);
}

// ------------------------------------------------------------------------------------------------
// stage: unified head (+ mask branch / controllers)   (unicorn_head.py:249-336, unicorn_head_mask.py:280-372)
// ------------------------------------------------------------------------------------------------
// K > 1 = object-batched call (row N3, unicorn_vos.py:178-200): ONE image (B == 1) with K prior sets.  The prior only enters at
// x = stem(fpn) + prior * beta (unicorn_head.py:272-277), so the FPN casts, the stem convs (+ their GroupNorm statistics) and the
// whole mask branch run once; prior fusion, attention blocks, towers, predictions and controllers run over K "samples".
int engine_head(uni_ctx* c, const float* fpn0, const float* fpn1, const float* fpn2, const float* prior8,
                const float* prior16, const float* prior32, int B, int K, int H, int W, int mode, float* out, float* dyn_params,
                float* mask_feats, float* up_masks, hipStream_t s) {
    const bool raw = (mode & 2) != 0;     // decode_in_inference = False (unicorn_head.py:436-439): [reg, sigmoid(obj), sigmoid(cls)] rows undecoded
    mode &= 1;
    UNI_REQUIRE(mode == 0 || mode == 1, "head: mode has to be 0 ('sot') or 1 ('mot')");   // unicorn_head.py:291-292
    UNI_REQUIRE(H % 32 == 0 && W % 32 == 0, "head: H=%d W=%d", H, W);
    UNI_REQUIRE(B >= 1 && B <= 64 && K >= 1 && K <= 64 && (K == 1 || B == 1), "head: batch %d x %d objects unsupported (B 1..64; K 1..64 with B = 1)", B, K);
    const int Bi = B;                      // images
    B = Bi * K;                            // samples of the per-object part
    RUN(stage_begin(c, B, H, W, s));
    c->nb = Bi;
    const auto& cfg = c->cfg;
    if (cfg.mask) UNI_REQUIRE(dyn_params && mask_feats && up_masks, "head: mask model needs dyn_params/mask_feats/up_masks");
    const int ch[3] = {cfg.dims[1], cfg.dims[2], cfg.dims[3]};
    const int Hk[3] = {H / 8, H / 16, H / 32}, Wk[3] = {W / 8, W / 16, W / 32};
    const int HWk[3] = {Hk[0] * Wk[0], Hk[1] * Wk[1], Hk[2] * Wk[2]};
    const int A = HWk[0] + HWk[1] + HWk[2];                       // anchors per image
    const float* fpn[3] = {fpn0, fpn1, fpn2};
    const float* prior[3] = {prior8, prior16, prior32};
    const int ncls = mode == 0 ? 1 : cfg.num_classes, nch = 5 + ncls;
    ActPtr fb[3];
    for (int k = 0; k < 3; ++k) {
        fb[k] = actalloc(c, (size_t)Bi * HWk[k] * ch[k]);
        RUN(prof_run(c, PC_MISC, 0.0, s, [&] { return launch_cast_operand(fpn[k], ch[k], fb[k], ch[k], Bi * HWk[k], ch[k], s, c->b32); }));
    }
    // The three FPN levels are independent until the decode: run them concurrently (the stride-16/32 levels are
    // far too small to fill 256 CUs on their own).  Level k gets its own stream and a disjoint workspace slice.
    static const bool no_fork = getenv("UNI_NO_FORK") != nullptr;     // A/B switch
    const bool fork = c->aux[0] && c->aux[1] && c->aux[2] && !c->prof_on && !no_fork;   // serialised while event-profiling so per-kernel times are not inflated by overlap
    static const bool no_fork_mb = getenv("UNI_NO_FORK_MB") != nullptr;                 // A/B switch: mask branch after the levels, on the caller's stream
    const bool fork_mb = fork && cfg.mask && !no_fork_mb;
    const int njoin = fork_mb ? 3 : 2;
    hipStream_t s_main = s;
    // an error inside the forked region must not leave the auxiliary streams unjoined (the caller's stream would otherwise race the
    // levels still in flight): RUN joins them before it returns
#undef RUN
#define RUN(expr)                                                                              \
    do {                                                                                       \
        int _rc = (expr);                                                                      \
        if (_rc || c->ws_overflow) {                                                           \
            if (fork)                                                                          \
                for (int _i = 0; _i < njoin; ++_i)                                             \
                    if (hipEventRecord(c->ev_join[_i], c->aux[_i]) == hipSuccess) (void)hipStreamWaitEvent(s_main, c->ev_join[_i], 0); \
            return _rc ? _rc : -3;                                                             \
        }                                                                                      \
    } while (0)
    auto hip_rc = [](hipError_t e, const char* what) {      // HIP failures inside the forked region go through RUN (which joins the aux streams)
        if (e == hipSuccess) return 0;
        uni_set_error("%s -> %s", what, hipGetErrorString(e));
        return -2;
    };
    if (fork) {
        RUN(hip_rc(hipEventRecord(c->ev_fork, s_main), "hipEventRecord(fork)"));
        for (int i = 0; i < njoin; ++i) RUN(hip_rc(hipStreamWaitEvent(c->aux[i], c->ev_fork, 0), "hipStreamWaitEvent(fork)"));
    }
    // condinst/mask_branch.py:77-99,158-162: depends on the FPN maps only -> once per IMAGE, and (forked) on its own stream BESIDE the three
    // levels: its GEMMs are 16000-row x 128-column problems (125 tiles of 128 x 128 at one frame) that leave half of the chip idle when they
    // run alone after the join (mask head 2.10 vs 1.49 ms for the plain head at one frame)
    auto mask_branch = [&](hipStream_t s) -> int {
        c->nb = Bi;
        const int M8 = Bi * HWk[0];
        float* xm = wsalloc<float>(c, (size_t)M8 * 128);
        for (int k = 0; k < 3; ++k) {
            const int M = Bi * HWk[k];
            float* r = k == 0 ? xm : wsalloc<float>(c, (size_t)M * 128);
            Out o; o.F = r; o.ldf = 128;
            RUN(run_conv_gn(c, c->refine[k], c->refine_gn[k], 16, 1e-3f, ACT_RELU, fb[k], ch[k], Hk[k], Wk[k], 1, o, s));
            if (k > 0) RUN(prof_run(c, PC_MISC, 0.0, s, [&] { return launch_add_aligned_bilinear(r, Hk[k], Wk[k], 128, Hk[0] / Hk[k], xm, s, Bi); }));
        }
        ActPtr xmb = actalloc(c, (size_t)M8 * 128);
        RUN(prof_run(c, PC_MISC, 0.0, s, [&] { return launch_cast_operand(xm, 128, xmb, 128, M8, 128, s, c->b32); }));
        ActPtr tb[2] = {actalloc(c, (size_t)M8 * 128), actalloc(c, (size_t)M8 * 128)};
        ActPtr cur = xmb;
        for (int i = 0; i < 4; ++i) {
            Out o; o.B = tb[i & 1]; o.ldb = 128;
            RUN(run_conv_gn(c, c->mtower[i], c->mtower_gn[i], 16, 1e-3f, ACT_RELU, cur, 128, Hk[0], Wk[0], 1, o, s));
            cur = tb[i & 1];
        }
        { GemmArgs g = conv_args(c->mtower_out, cur, 128, M8, 1, 1, 0); g.outF = mask_feats; g.ldf = 8; RUN(p_gemm(c, g, s)); }
        ActPtr u = tb[0] == cur ? tb[1] : tb[0];
        { GemmArgs g = conv_args(c->upm0, xmb, 128, Hk[0], Wk[0], 1, 1, Bi); g.act = ACT_RELU; g.outB = u; g.ldb = 128; RUN(p_gemm(c, g, s)); }
        { GemmArgs g = conv_args(c->upm1, u, 128, M8, 1, 1, 0); g.outF = up_masks; g.ldf = c->upm1.N; RUN(p_gemm(c, g, s)); }
        return 0;
    };
    const size_t lvl_base = (c->ws_off + 255) & ~(size_t)255;
    const size_t M0 = (size_t)B * HWk[0];
    const size_t slice_bytes = ((M0 * 16384) + 65536 + UNI_SLAB_BYTES + 4096 + 255) & ~(size_t)255;   // >= per-level footprint at level 0 (bf16 9.2 KB/pixel, fp32 15.4 KB/pixel) + one split-K slab
    const int row_start[3] = {0, HWk[0], HWk[0] + HWk[1]};
    size_t mb_end = 0;
    if (fork_mb) {      // its workspace lies behind the three level slices (where the serial version allocates it, too)
        c->ws_off = lvl_base + 3 * slice_bytes;
        RUN(mask_branch(c->aux[2]));
        mb_end = c->ws_off;
    }
    for (int k = 0; k < 3; ++k) {
        const int M = B * HWk[k];                                // rows over the batch at this level
        const int row0 = row_start[k];
        if (fork) { s = k == 0 ? s_main : c->aux[k - 1]; c->ws_off = lvl_base + (size_t)k * slice_bytes; }
        const size_t mark = c->ws_off;
        float* x = wsalloc<float>(c, (size_t)M * 256);
        if (K == 1) {
            c->nb = Bi;
            Out o; o.F = x; o.ldf = 256; o.prior = prior[k]; o.pbeta = c->beta[k];
            RUN(run_baseconv(c, c->stems[k], fb[k], ch[k], Hk[k], Wk[k], o, s));
        } else {   // stem conv + statistics once, GroupNorm-apply + SiLU + prior fusion once per object
            c->nb = 1;
            const size_t mk = c->ws_off;
            GemmArgs g = conv_args(c->stems[k].conv, fb[k], ch[k], Hk[k], Wk[k], 1, 0, 1);
            float* raw = wsalloc<float>(c, (size_t)g.M * g.N);
            double* st = next_stats(c, 1);
            g.outF = raw; g.ldf = g.N; g.stats = st; g.cpg = g.N / 16;
            RUN(p_gemm(c, g, s));
            for (int ko = 0; ko < K; ++ko) {
                GnApplyArgs a;
                a.x = raw; a.ldx = g.N; a.stats = st; a.gamma = c->stems[k].gn.g; a.beta = c->stems[k].gn.b; a.eps = 1e-3f;
                a.M = HWk[k]; a.B = 1; a.C = 256; a.G = 16; a.act = ACT_SILU;
                a.prior = prior[k] + (size_t)ko * HWk[k]; a.prior_beta = c->beta[k];
                a.outF = x + (size_t)ko * HWk[k] * 256; a.ldf = 256; a.W = Wk[k];
                RUN(p_gn(c, a, s));
            }
            (void)mk;       // raw stays allocated until the level's slice is released (the gn_apply launches read it)
        }
        c->nb = B;
        ActPtr t = actalloc(c, (size_t)M * 256);
        ActPtr hid = actalloc(c, (size_t)M * 1024);
        ActPtr xb = actalloc(c, (size_t)M * 256);
        const int nblk = (int)c->att[k].size();
        if (nblk == 0) RUN(prof_run(c, PC_MISC, 0.0, s, [&] { return launch_cast_operand(x, 256, xb, 256, M, 256, s, c->b32); }));
        for (int n = 0; n < nblk; ++n) RUN(run_block(c, c->att[k][n], x, Hk[k], Wk[k], t, hid, n == nblk - 1 ? xb : ActPtr(), s));
        ActPtr tw = actalloc(c, (size_t)M * 512);      // [cls | reg] after the first (merged) tower conv
        { Out o; o.B = tw; o.ldb = 512; RUN(run_conv_gn(c, c->tower0[k], c->tower0_gn[k], 32, 1e-3f, ACT_SILU, xb, 256, Hk[k], Wk[k], 1, o, s)); }
        ActPtr cb[2] = {actalloc(c, (size_t)M * 256), actalloc(c, (size_t)M * 256)};
        ActPtr rb[2] = {actalloc(c, (size_t)M * 256), actalloc(c, (size_t)M * 256)};
        ActPtr cls = tw, reg = tw + 256; int ldc = 512;
        for (int i = 1; i < 4; ++i) {
            { Out o; o.B = cb[i & 1]; o.ldb = 256; RUN(run_baseconv(c, c->cls_convs[k][i], cls, ldc, Hk[k], Wk[k], o, s)); }
            { Out o; o.B = rb[i & 1]; o.ldb = 256; RUN(run_baseconv(c, c->reg_convs[k][i], reg, ldc, Hk[k], Wk[k], o, s)); }
            cls = cb[i & 1]; reg = rb[i & 1]; ldc = 256;
        }
        // predictions land in out[b][row0 + p][:] (anchor order: level 8 rows, then 16, then 32, per image)
        {   // [reg(4) | sigmoid(obj)]  (unicorn_head.py:295-304,332-334)
            GemmArgs g = conv_args(mode == 0 ? c->regobj_sot[k] : c->regobj[k], reg, ldc, M, 1, 1, 0);
            g.act = ACT_SIGMOID; g.act_col0 = 4; g.outF = out; g.ldf = nch;
            g.out_hw = HWk[k]; g.out_stride = A; g.out_off = row0;
            RUN(p_gemm(c, g, s));
        }
        {   // sigmoid(cls)
            GemmArgs g = conv_args(mode == 0 ? c->cls_pred_sot[k] : c->cls_pred[k], cls, ldc, M, 1, 1, 0);
            g.act = ACT_SIGMOID; g.outF = out + 5; g.ldf = nch;
            g.out_hw = HWk[k]; g.out_stride = A; g.out_off = row0;
            RUN(p_gemm(c, g, s));
        }
        if (cfg.mask) {   // controllers on reg_feat (ctrl_loc == "reg", unicorn_head_mask.py:333-340)
            GemmArgs g = conv_args(c->controllers[k], reg, ldc, Hk[k], Wk[k], 1, 1, B);
            g.outF = dyn_params; g.ldf = 169;
            g.out_hw = HWk[k]; g.out_stride = A; g.out_off = row0;
            RUN(p_gemm(c, g, s));
        }
        if (!fork) c->ws_off = mark;
    }
    s = s_main;
    if (fork) {
        for (int i = 0; i < njoin; ++i) {
            RUN(hip_rc(hipEventRecord(c->ev_join[i], c->aux[i]), "hipEventRecord(join)"));
            RUN(hip_rc(hipStreamWaitEvent(s_main, c->ev_join[i], 0), "hipStreamWaitEvent(join)"));
        }
        c->ws_off = fork_mb ? mb_end : lvl_base + 3 * slice_bytes;
    }
#undef RUN
#define RUN(expr) do { int _rc = (expr); if (_rc) return _rc; if (c->ws_overflow) return -3; } while (0)
    if (!raw) RUN(prof_run(c, PC_MISC, 0.0, s, [&] { return launch_decode(out, out, HWk[0], Wk[0], HWk[1], Wk[1], HWk[2], Wk[2], nch, s, B); }));
    if (cfg.mask && !fork_mb) RUN(mask_branch(s));
    return 0;
}

void engine_destroy(uni_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (void* p : c->dev_allocs) (void)hipFree(p);
    if (c->ws) (void)hipFree(c->ws);
    for (int i = 0; i < 3; ++i) { if (c->aux[i]) (void)hipStreamDestroy(c->aux[i]); if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    delete c;
}
