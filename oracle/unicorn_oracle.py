"""CPU oracle for Unicorn's per-frame inference hot path.

TEST INFRASTRUCTURE ONLY — a plain fp32 PyTorch-CPU restatement of the reference algorithm,
written as pure functions over a flat ``state_dict`` (same key names as the reference checkpoint).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it;
the product package ``unicorn_amd`` never does (it fails loudly when the HIP library is missing).

Parity pin: ``tests/golden/*.npz`` were produced by the REAL reference modules imported in the
build container (``tests/golden/make_golden.py`` via ``oracle/ref_bootstrap.py``) with the
synthetic weights of ``oracle/synth.py``; ``tests/test_oracle_golden.py`` checks this file against
them (re-generate with ``python tests/golden/make_golden.py`` wherever ``/root/reference`` is present: the arrays are
bit-reproducible).  Third-party arithmetic not under /root/reference
(torchvision nms/batched_nms, pinned 0.11.x by assets/install.md:8,14) is restated from its
published semantics; the reference has no test pinning NMS results -> that part is "parity
unpinned" (SURVEY.md §8c).

Every function cites the reference file:line it follows (paths relative to /root/reference).
All tensors are NCHW fp32 like the reference's.
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import math
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass
class ModelCfg:
    """Subset of exp/unicorn_track.py:31-113 + exps/default/*.py that shapes the inference net."""
    name: str = "unicorn_track_tiny"
    dims: Tuple[int, ...] = (96, 192, 384, 768)          # convnext.py:198-211
    depths: Tuple[int, ...] = (3, 3, 9, 3)
    num_classes: int = 8                                  # exp/unicorn_track.py:37
    mask: bool = False                                    # ExpTrackMask (unicorn_track_mask.py)
    n_layer_att: int = 3
    embed_dim: int = 128
    hidden: int = 256
    up_rate: int = 4                                      # 8 // d_rate, unicorn_track_mask.py:44,64
    d_rate: int = 2

    @property
    def in_channels(self):
        return tuple(self.dims[1:])


CONFIGS = {
    "unicorn_track_tiny": ModelCfg("unicorn_track_tiny"),
    "unicorn_track_tiny_mask": ModelCfg("unicorn_track_tiny_mask", mask=True),
    "unicorn_track_large": ModelCfg("unicorn_track_large", dims=(192, 384, 768, 1536), depths=(3, 3, 27, 3)),
    "unicorn_track_large_mask": ModelCfg("unicorn_track_large_mask", dims=(192, 384, 768, 1536),
                                         depths=(3, 3, 27, 3), mask=True),
    # exps/default/unicorn_track_large_mot_challenge.py:18 -> num_classes = 1
    "unicorn_track_large_mot_challenge": ModelCfg("unicorn_track_large_mot_challenge",
                                                  dims=(192, 384, 768, 1536), depths=(3, 3, 27, 3),
                                                  num_classes=1),
    "unicorn_track_large_mot_challenge_mask": ModelCfg("unicorn_track_large_mot_challenge_mask",
                                                       dims=(192, 384, 768, 1536), depths=(3, 3, 27, 3),
                                                       num_classes=1, mask=True),
}


# --------------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------------
def ln_channels_first(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-6) -> Tensor:
    """convnext.py:179-184 (biased variance over C per pixel, eps inside sqrt)."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return w[:, None, None] * x + b[:, None, None]


def convnext_block(P: Dict[str, Tensor], p: str, x: Tensor) -> Tensor:
    """convnext.py:41-54: dw7x7(+bias) -> LN(C,1e-6) -> Linear C->4C -> GELU(erf) -> Linear -> gamma -> +res."""
    C = x.shape[1]
    y = F.conv2d(x, P[p + "dwconv.weight"], P[p + "dwconv.bias"], padding=3, groups=C)
    y = y.permute(0, 2, 3, 1)
    y = F.layer_norm(y, (C,), P[p + "norm.weight"], P[p + "norm.bias"], 1e-6)
    y = F.linear(y, P[p + "pwconv1.weight"], P[p + "pwconv1.bias"])
    y = F.gelu(y)
    y = F.linear(y, P[p + "pwconv2.weight"], P[p + "pwconv2.bias"])
    y = P[p + "gamma"] * y
    return x + y.permute(0, 3, 1, 2)


def convnext_features(P, cfg: ModelCfg, img: Tensor, p: str = "backbone.backbone.") -> List[Tensor]:
    """convnext.py:141-154 with out_indices [1,2,3]."""
    outs = []
    x = img
    for i in range(4):
        d = p + "downsample_layers.%d." % i
        if i == 0:
            x = F.conv2d(x, P[d + "0.weight"], P[d + "0.bias"], stride=4)
            x = ln_channels_first(x, P[d + "1.weight"], P[d + "1.bias"])
        else:
            x = ln_channels_first(x, P[d + "0.weight"], P[d + "0.bias"])
            x = F.conv2d(x, P[d + "1.weight"], P[d + "1.bias"], stride=2)
        for j in range(cfg.depths[i]):
            x = convnext_block(P, p + "stages.%d.%d." % (i, j), x)
        if i >= 1:
            outs.append(ln_channels_first(x, P[p + "norm%d.weight" % i], P[p + "norm%d.bias" % i]))
    return outs  # [x2 (s8), x1 (s16), x0 (s32)]


def base_conv(P, p: str, x: Tensor, k: int, stride: int = 1) -> Tensor:
    """network_blocks.py:29-51 after BN->GN conversion (exp/unicorn_track.py:450-470):
    conv(no bias) -> GroupNorm(16, eps=1e-3) -> SiLU."""
    y = F.conv2d(x, P[p + "conv.weight"], None, stride=stride, padding=(k - 1) // 2)
    y = F.group_norm(y, 16, P[p + "bn.weight"], P[p + "bn.bias"], 1e-3)
    return F.silu(y)


def csp_layer(P, p: str, x: Tensor, n: int = 3) -> Tensor:
    """network_blocks.py:180-185, Bottleneck :97-101 with shortcut=False (yolo_pafpn_new.py:70)."""
    x1 = base_conv(P, p + "conv1.", x, 1)
    x2 = base_conv(P, p + "conv2.", x, 1)
    for i in range(n):
        x1 = base_conv(P, p + "m.%d.conv2." % i, base_conv(P, p + "m.%d.conv1." % i, x1, 1), 3)
    return base_conv(P, p + "conv3.", torch.cat((x1, x2), 1), 1)


def pafpn(P, feats: List[Tensor], p: str = "backbone.") -> Tuple[Tensor, Tensor, Tensor]:
    """yolo_pafpn_new.py:113-161 (width == 1 -> no adjust convs)."""
    x2, x1, x0 = feats
    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
    fpn_out0 = base_conv(P, p + "lateral_conv0.", x0, 1)
    f_out0 = csp_layer(P, p + "C3_p4.", torch.cat([up(fpn_out0), x1], 1))
    fpn_out1 = base_conv(P, p + "reduce_conv1.", f_out0, 1)
    pan_out2 = csp_layer(P, p + "C3_p3.", torch.cat([up(fpn_out1), x2], 1))
    p_out1 = base_conv(P, p + "bu_conv2.", pan_out2, 3, 2)
    pan_out1 = csp_layer(P, p + "C3_n3.", torch.cat([p_out1, fpn_out1], 1))
    p_out0 = base_conv(P, p + "bu_conv1.", pan_out1, 3, 2)
    pan_out0 = csp_layer(P, p + "C3_n4.", torch.cat([p_out0, fpn_out0], 1))
    return pan_out2, pan_out1, pan_out0


def pos_embed(P, h: int, w: int) -> Tensor:
    """position_encoding.py:25-36 + unicorn.py:248-250 (same-size bicubic resample == identity)."""
    row, col = P["pos_emb.row_embed.weight"], P["pos_emb.col_embed.weight"]
    sz = row.shape[0]
    pos = torch.cat([col.unsqueeze(0).repeat(sz, 1, 1), row.unsqueeze(1).repeat(1, sz, 1)], dim=-1)
    pos = pos.permute(2, 0, 1).unsqueeze(0)
    pos = F.interpolate(pos, (h, w), mode="bilinear", align_corners=False)
    return F.interpolate(pos, size=(h, w), mode="bicubic")


def forward_backbone(P, cfg: ModelCfg, img: Tensor):
    """unicorn.py:231-258 with run_fpn=True."""
    feats = convnext_features(P, cfg, img)
    fpn = pafpn(P, feats)
    feat16 = feats[1]
    h, w = feat16.shape[-2:]
    return fpn, {"feat": feat16, "pos": pos_embed(P, h, w), "h": h, "w": w}


# --------------------------------------------------------------------------------------------
# deformable interaction
# --------------------------------------------------------------------------------------------
def msda_core(value: Tensor, shapes: List[Tuple[int, int]], loc: Tensor, attn: Tensor) -> Tensor:
    """Bilinear multi-scale sampling + attention-weighted reduction, restating the CUDA forward
    kernel ms_deform_im2col_cuda.cuh:237-299 (bilinear :33-84): pixel coords = loc*(W,H) - 0.5, sample
    skipped unless h_im > -1 and w_im > -1 and h_im < H and w_im < W, corners outside the map add 0.
    value (N,S,M,D); loc (N,Lq,M,L,P,2) normalised (x,y); attn (N,Lq,M,L,P).  Returns (N,Lq,M*D)."""
    N, S, M, D = value.shape
    _, Lq, _, L, Pn, _ = loc.shape
    out = torch.zeros(N, Lq, M, D, dtype=value.dtype)
    start = 0
    for l, (H, W) in enumerate(shapes):
        v = value[:, start:start + H * W]                       # (N, HW, M, D)
        start += H * W
        x = loc[:, :, :, l, :, 0] * W - 0.5                     # (N, Lq, M, P)
        y = loc[:, :, :, l, :, 1] * H - 0.5
        valid = (y > -1) & (x > -1) & (y < H) & (x < W)
        x0 = torch.floor(x)
        y0 = torch.floor(y)
        lx, ly = x - x0, y - y0
        x0, y0 = x0.long(), y0.long()
        acc = torch.zeros(N, Lq, M, Pn, D, dtype=value.dtype)
        for dy, dx, wgt in ((0, 0, (1 - ly) * (1 - lx)), (0, 1, (1 - ly) * lx),
                            (1, 0, ly * (1 - lx)), (1, 1, ly * lx)):
            yy, xx = y0 + dy, x0 + dx
            ok = valid & (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
            idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1))  # (N,Lq,M,P)
            vm = v.permute(0, 2, 1, 3)                           # (N, M, HW, D); gather v[n, idx, m, :]
            g = torch.gather(vm, 2, idx.permute(0, 2, 1, 3).reshape(N, M, Lq * Pn, 1).expand(N, M, Lq * Pn, D))
            g = g.reshape(N, M, Lq, Pn, D).permute(0, 2, 1, 3, 4)
            acc = acc + g * (wgt * ok)[..., None]
        out = out + (acc * attn[:, :, :, l, :, None]).sum(3)
    return out.reshape(N, Lq, M * D)


def msda_module(P, p: str, query: Tensor, ref_pts: Tensor, src: Tensor, shapes) -> Tensor:
    """ops/modules/ms_deform_attn.py:78-115 (n_heads 8, n_levels 2, n_points 4; no padding mask)."""
    N, Lq, C = query.shape
    M, L, Pn = 8, len(shapes), 4
    value = F.linear(src, P[p + "value_proj.weight"], P[p + "value_proj.bias"]).view(N, -1, M, C // M)
    off = F.linear(query, P[p + "sampling_offsets.weight"], P[p + "sampling_offsets.bias"]).view(N, Lq, M, L, Pn, 2)
    aw = F.linear(query, P[p + "attention_weights.weight"], P[p + "attention_weights.bias"]).view(N, Lq, M, L * Pn)
    aw = F.softmax(aw, -1).view(N, Lq, M, L, Pn)
    norm = torch.tensor([[w, h] for (h, w) in shapes], dtype=query.dtype)
    loc = ref_pts[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    out = msda_core(value, shapes, loc, aw)
    return F.linear(out, P[p + "output_proj.weight"], P[p + "output_proj.bias"])


def deform_transformer(P, srcs: List[Tensor], poss: List[Tensor], p: str = "transformer.") -> Tensor:
    """deformable_transformer.py:58-89 (+ encoder :141-163, layer :122-131); eval => dropout off;
    masks all False => valid_ratios == 1."""
    shapes = [tuple(s.shape[-2:]) for s in srcs]
    src = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
    pos = torch.cat([ps.flatten(2).transpose(1, 2) + P[p + "level_embed"][l].view(1, 1, -1)
                     for l, ps in enumerate(poss)], 1)
    refs = []
    for (H, W) in shapes:
        ry, rx = torch.meshgrid(torch.linspace(0.5, H - 0.5, H), torch.linspace(0.5, W - 0.5, W), indexing="ij")
        refs.append(torch.stack((rx.reshape(-1) / W, ry.reshape(-1) / H), -1))
    ref = torch.cat(refs, 0)[None, :, None, :].expand(src.shape[0], -1, len(shapes), -1)
    e = p + "encoder.layers.0."
    src2 = msda_module(P, e + "self_attn.", src + pos, ref, src, shapes)
    src = F.layer_norm(src + src2, (src.shape[-1],), P[e + "norm1.weight"], P[e + "norm1.bias"], 1e-5)
    src2 = F.linear(F.relu(F.linear(src, P[e + "linear1.weight"], P[e + "linear1.bias"])),
                    P[e + "linear2.weight"], P[e + "linear2.bias"])
    return F.layer_norm(src + src2, (src.shape[-1],), P[e + "norm2.weight"], P[e + "norm2.bias"], 1e-5)


def bottleneck(P, x: Tensor) -> Tensor:
    """unicorn.py:36-38: 1x1 conv (+bias) -> GroupNorm(32, eps 1e-5)."""
    y = F.conv2d(x, P["bottleneck.0.weight"], P["bottleneck.0.bias"])
    return F.group_norm(y, 32, P["bottleneck.1.weight"], P["bottleneck.1.bias"], 1e-5)


def forward_interaction(P, d0: dict, d1: dict) -> Tuple[Tensor, Tensor]:
    """unicorn.py:260-276."""
    srcs = [bottleneck(P, d0["feat"]), bottleneck(P, d1["feat"])]
    out = deform_transformer(P, srcs, [d0["pos"], d1["pos"]])
    bs, n, c = out.shape
    h, w = d0["h"], d0["w"]
    a, b = out[:, :n // 2], out[:, n // 2:]
    return a.permute(0, 2, 1).reshape(bs, c, h, w), b.permute(0, 2, 1).reshape(bs, c, h, w)


def forward_upsample(P, x: Tensor) -> Tensor:
    """unicorn.py:41-44,311-313: PixelShuffle(2) -> conv3x3+bias -> ReLU -> conv3x3+bias."""
    x = F.pixel_shuffle(x, 2)
    x = F.relu(F.conv2d(x, P["upsample_layer.1.weight"], P["upsample_layer.1.bias"], padding=1))
    return F.conv2d(x, P["upsample_layer.3.weight"], P["upsample_layer.3.bias"], padding=1)


# --------------------------------------------------------------------------------------------
# dense correlation / propagation (SOT / VOS drivers)
# --------------------------------------------------------------------------------------------
def correlation_propagate(embed_ref: Tensor, embed_cur: Tensor, values: Tensor, half: bool = False) -> Tensor:
    """external/lib/test/tracker/unicorn_sot.py:88-100 (same in unicorn_vos.py:166-186):
    simi = E_ref^T E_cur (HW x HW), softmax over the REFERENCE axis (dim 0), pred = values @ trans.
    embed_*: (C, HW); values: (K, HW_ref) -> (K, HW_cur).  `half=True` emulates the driver's fp16 casts
    (:95-97); default is the fp32 definition used for parity (SURVEY.md §7 hard parts)."""
    if half:
        k, q, v = embed_ref.half().float(), embed_cur.half().float(), values.half().float()
    else:
        k, q, v = embed_ref, embed_cur, values
    out = torch.empty(v.shape[0], q.shape[1], dtype=torch.float32)
    step = 2048  # column blocks: never materialise more than HW x 2048
    for s in range(0, q.shape[1], step):
        simi = k.t() @ q[:, s:s + step]
        if half:
            simi = simi.half().float()
        trans = torch.softmax(simi, dim=0)
        if half:
            trans = trans.half().float()
        out[:, s:s + step] = v @ trans
    return out


def prior_pyramid(coarse: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """unicorn_sot.py:103-105."""
    return (coarse, F.interpolate(coarse, scale_factor=1 / 2, mode="bilinear", align_corners=False),
            F.interpolate(coarse, scale_factor=1 / 4, mode="bilinear", align_corners=False))


def get_label_map(box_xyxy: Tensor, H: int, W: int) -> Tensor:
    """unicorn_sot.py:128-139."""
    lab = torch.zeros((1, 1, H, W), dtype=torch.float32)
    x1, y1, x2, y2 = torch.round(box_xyxy).int().tolist()
    x1, x2 = max(0, min(x1, W)), max(0, min(x2, W))
    y1, y2 = max(0, min(y1, H)), max(0, min(y2, H))
    lab[0, 0, y1:y2, x1:x2] = 1.0
    return lab


def label_map_s8(box_xyxy: Tensor, H: int, W: int) -> Tensor:
    """unicorn_sot.py:52-53 -> (1, H/8*W/8)."""
    return F.interpolate(get_label_map(box_xyxy, H, W), scale_factor=1 / 8, mode="bilinear",
                         align_corners=False)[0].flatten(-2)


# --------------------------------------------------------------------------------------------
# heads
# --------------------------------------------------------------------------------------------
def _head_trunk(P, cfg: ModelCfg, fpn, priors, mode: str):
    """unicorn_head.py:266-334 / unicorn_head_mask.py:294-372 (inference branch)."""
    assert mode in ("sot", "mot")
    outs, dyn = [], []
    for k, (x, m) in enumerate(zip(fpn, priors)):
        x = base_conv(P, "head.stems.%d." % k, x, 1)
        x = x + m * P["head.beta_%d" % k]                       # learnable_fuse, beta index = level
        for n in range(cfg.n_layer_att):
            x = convnext_block(P, "head.att_layers.%d.%d." % (k, n), x)
        cls_feat, reg_feat = x, x
        for i in range(4):
            cls_feat = base_conv(P, "head.cls_convs.%d.%d." % (k, i), cls_feat, 3)
            reg_feat = base_conv(P, "head.reg_convs.%d.%d." % (k, i), reg_feat, 3)
        sfx = "_sot" if mode == "sot" else ""
        cls_o = F.conv2d(cls_feat, P["head.cls_preds%s.%d.weight" % (sfx, k)], P["head.cls_preds%s.%d.bias" % (sfx, k)])
        reg_o = F.conv2d(reg_feat, P["head.reg_preds%s.%d.weight" % (sfx, k)], P["head.reg_preds%s.%d.bias" % (sfx, k)])
        obj_o = F.conv2d(reg_feat, P["head.obj_preds%s.%d.weight" % (sfx, k)], P["head.obj_preds%s.%d.bias" % (sfx, k)])
        outs.append(torch.cat([reg_o, obj_o.sigmoid(), cls_o.sigmoid()], 1))
        if cfg.mask:  # ctrl_loc == "reg" (unicorn_track_mask.py:38)
            d = F.conv2d(reg_feat, P["head.controllers.%d.weight" % k], P["head.controllers.%d.bias" % k], padding=1)
            dyn.append(d.flatten(-2).permute(0, 2, 1))
    return outs, dyn


def decode_outputs(outs: List[Tensor], strides=(8, 16, 32)):
    """unicorn_head.py:430-439,467-482 / unicorn_head_mask.py:502-519."""
    hw = [o.shape[-2:] for o in outs]
    out = torch.cat([o.flatten(2) for o in outs], 2).permute(0, 2, 1).contiguous()
    grids, strs = [], []
    for (h, w), s in zip(hw, strides):
        yv, xv = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        grids.append(torch.stack((xv, yv), 2).view(1, -1, 2).float())
        strs.append(torch.full((1, h * w, 1), float(s)))
    grids, strs = torch.cat(grids, 1), torch.cat(strs, 1)
    out[..., :2] = (out[..., :2] + grids) * strs
    out[..., 2:4] = torch.exp(out[..., 2:4]) * strs
    locations = ((grids + 0.5) * strs)[0]
    return out, locations


def head_forward(P, cfg: ModelCfg, fpn, priors, mode: str) -> Tensor:
    """UnicornHead.forward inference path -> (1, n_anchors, 5+nc) decoded."""
    outs, _ = _head_trunk(P, cfg, fpn, priors, mode)
    return decode_outputs(outs)[0]


def aligned_bilinear(t: Tensor, factor: int) -> Tensor:
    """condinst/comm.py:5-27."""
    if factor == 1:
        return t
    h, w = t.shape[2:]
    t = F.pad(t, pad=(0, 1, 0, 1), mode="replicate")
    oh, ow = factor * h + 1, factor * w + 1
    t = F.interpolate(t, size=(oh, ow), mode="bilinear", align_corners=True)
    t = F.pad(t, pad=(factor // 2, 0, factor // 2, 0), mode="replicate")
    return t[:, :, :oh - 1, :ow - 1]


def _mb_conv(P, p: str, x: Tensor) -> Tensor:
    """conv_with_kaiming_uniform.py: conv3x3 (no bias) -> BN(->GN16, eps 1e-3) -> ReLU."""
    y = F.conv2d(x, P[p + "0.weight"], None, padding=1)
    return F.relu(F.group_norm(y, 16, P[p + "1.weight"], P[p + "1.bias"], 1e-3))


def mask_branch(P, cfg: ModelCfg, fpn) -> Tuple[Tensor, Tensor]:
    """condinst/mask_branch.py:77-99,158-162 (use_raft=True)."""
    p = "head.mask_branch."
    x = _mb_conv(P, p + "refine.0.", fpn[0])
    for i in (1, 2):
        xp = _mb_conv(P, p + "refine.%d." % i, fpn[i])
        x = x + aligned_bilinear(xp, x.shape[2] // xp.shape[2])
    t = x
    for i in range(4):
        t = _mb_conv(P, p + "tower.%d." % i, t)
    mask_feats = F.conv2d(t, P[p + "tower.4.weight"], P[p + "tower.4.bias"])
    u = F.relu(F.conv2d(x, P[p + "up_mask_layer.0.weight"], P[p + "up_mask_layer.0.bias"], padding=1))
    up_masks = F.conv2d(u, P[p + "up_mask_layer.2.weight"], P[p + "up_mask_layer.2.bias"])
    return mask_feats, up_masks


def head_mask_forward(P, cfg: ModelCfg, fpn, priors, mode: str):
    """UnicornHeadMask.forward inference path (unicorn_head_mask.py:451-471) -> 6-tuple."""
    outs, dyn = _head_trunk(P, cfg, fpn, priors, mode)
    mask_feats, up_masks = mask_branch(P, cfg, fpn)
    out, locations = decode_outputs(outs)
    levels = torch.cat([torch.full((1, o.shape[-2] * o.shape[-1]), k) for k, o in enumerate(outs)], 1)
    return out, locations, torch.cat(dyn, 1), levels, mask_feats, up_masks


SOI = (64.0, 128.0, 256.0, 512.0, 1024.0)  # dynamic_mask_head.py:107


def dynamic_mask_head(cfg: ModelCfg, mask_feats: Tensor, params: Tensor, inst_loc: Tensor,
                      inst_lvl: Tensor, up_masks: Tensor, stride: int = 8) -> Tensor:
    """condinst/dynamic_mask_head.py:172-225 (+ parse :61-87, heads :138-156, convex upsample :159-170)
    -> sigmoid mask scores (N,1,up_rate*H,up_rate*W)."""
    _, Cin, H, W = mask_feats.shape
    N = params.shape[0]
    ys, xs = torch.meshgrid(torch.arange(0, H * stride, stride, dtype=torch.float32),
                            torch.arange(0, W * stride, stride, dtype=torch.float32), indexing="ij")
    loc = torch.stack((xs.reshape(-1), ys.reshape(-1)), 1) + stride // 2           # comm.py:30-43
    rel = (inst_loc.reshape(-1, 1, 2) - loc.reshape(1, -1, 2)).permute(0, 2, 1).float()
    soi = torch.tensor(SOI)[inst_lvl.long()]
    rel = rel / soi.reshape(-1, 1, 1)
    x = torch.cat([rel, mask_feats[0].reshape(1, Cin, H * W).expand(N, -1, -1)], 1)  # (N,10,HW)
    ch = 8
    w0, w1, w2, b0, b1, b2 = torch.split(params, [(Cin + 2) * ch, ch * ch, ch, ch, ch, 1], 1)
    x = F.relu(torch.bmm(w0.reshape(N, ch, Cin + 2), x) + b0.reshape(N, ch, 1))
    x = F.relu(torch.bmm(w1.reshape(N, ch, ch), x) + b1.reshape(N, ch, 1))
    x = torch.bmm(w2.reshape(N, 1, ch), x) + b2.reshape(N, 1, 1)
    logits = x.reshape(N, 1, H, W)
    r = cfg.up_rate
    m = torch.softmax(up_masks.view(1, 1, 9, r, r, H, W), dim=2)
    up = F.unfold(logits, [3, 3], padding=1).view(N, 1, 9, 1, 1, H, W)
    up = torch.sum(m * up, dim=2).permute(0, 1, 4, 2, 5, 3).reshape(N, 1, r * H, r * W)
    return up.sigmoid()


# --------------------------------------------------------------------------------------------
# post-processing (torchvision semantics restated; "parity unpinned", see module docstring)
# --------------------------------------------------------------------------------------------
def nms(boxes: Tensor, scores: Tensor, thr: float) -> Tensor:
    """torchvision.ops.nms: greedy, descending score, suppress IoU > thr. Returns kept indices
    in descending-score order."""
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.long)
    order = torch.argsort(scores, descending=True, stable=True)
    b = boxes[order]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    sup = torch.zeros(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if sup[i]:
            continue
        keep.append(i)
        lt = torch.max(b[i, :2], b[i + 1:, :2])
        rb = torch.min(b[i, 2:], b[i + 1:, 2:])
        wh = (rb - lt).clamp(min=0)
        inter = wh[:, 0] * wh[:, 1]
        iou = inter / (area[i] + area[i + 1:] - inter)
        sup[i + 1:] |= iou > thr
    return order[torch.tensor(keep, dtype=torch.long)]


def batched_nms(boxes, scores, idxs, thr):
    """torchvision.ops.batched_nms (coordinate trick: offset = idx * (max_coordinate + 1))."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.long)
    off = idxs.to(boxes) * (boxes.max() + 1)
    return nms(boxes + off[:, None], scores, thr)


def _to_corners_(pred: Tensor) -> Tensor:
    c = pred.new_empty(pred.shape[:-1] + (4,))
    c[..., 0] = pred[..., 0] - pred[..., 2] / 2
    c[..., 1] = pred[..., 1] - pred[..., 3] / 2
    c[..., 2] = pred[..., 0] + pred[..., 2] / 2
    c[..., 3] = pred[..., 1] + pred[..., 3] / 2
    pred[..., :4] = c
    return pred


def postprocess(pred: Tensor, num_classes: int, conf_thre: float, nms_thre: float,
                class_agnostic: bool = False, return_index: bool = False):
    """utils/boxes.py:33-77. pred (1,A,5+nc) decoded cxcywh; mutated in place like the reference."""
    pred = _to_corners_(pred)
    outs = []
    for ip in pred:
        cc, cp = torch.max(ip[:, 5:5 + num_classes], 1, keepdim=True)
        mask = (ip[:, 4] * cc.squeeze(1) >= conf_thre)
        det = torch.cat((ip[:, :5], cc, cp.float()), 1)[mask]
        if det.shape[0] == 0:
            outs.append((None, None) if return_index else None)
            continue
        sc = det[:, 4] * det[:, 5]
        keep = nms(det[:, :4], sc, nms_thre) if class_agnostic else batched_nms(det[:, :4], sc, det[:, 6], nms_thre)
        if return_index:
            outs.append((det[keep], torch.nonzero(mask).squeeze(1)[keep]))
        else:
            outs.append(det[keep])
    return outs


def postprocess_inst(cfg: ModelCfg, head_out, num_classes, conf_thre, nms_thre, class_agnostic=False):
    """utils/boxes.py:80-152 for batch 1 -> (detections (M,7), masks (M,1,H,W)) or (None, None)."""
    pred, locations, dyn, levels, mask_feats, up_masks = head_out
    det, idx = postprocess(pred, num_classes, conf_thre, nms_thre, class_agnostic, return_index=True)[0]
    if det is None:
        return None, None
    m = dynamic_mask_head(cfg, mask_feats, dyn[0][idx], locations[idx], levels[0][idx], up_masks)
    return det, aligned_bilinear(m, cfg.d_rate)


def sample_instance_embeddings(embed: Tensor, boxes_xyxy: Tensor, stride: int = 8) -> Tensor:
    """evaluators/mot_evaluator.py:1024-1034 (also :822-827): bilinear grid_sample of the embedding map at box centres
    (border padding, align_corners=False).  The reference normalises the clamped stride-8 centre by (W8 - 1) and then
    samples with align_corners=False, i.e. at x = clamp(c/s - 0.5, 0, W8-1) * W8/(W8-1) - 0.5; restated literally
    (pinned by tests/golden/sample_embed_ref.npz, produced by exec-ing those reference lines).
    embed (1,C,H,W), boxes in input-image pixels -> (N,C)."""
    _, C, H, W = embed.shape
    s = stride
    cx, cy = (boxes_xyxy[:, 0] + boxes_xyxy[:, 2]) / 2 / s - 0.5, (boxes_xyxy[:, 1] + boxes_xyxy[:, 3]) / 2 / s - 0.5
    gx = (torch.clamp(cx, min=0, max=W - 1) / (W - 1) - 0.5) * 2.0
    gy = (torch.clamp(cy, min=0, max=H - 1) / (H - 1) - 0.5) * 2.0
    grid = torch.stack((gx, gy), -1).view(1, -1, 1, 2)
    return F.grid_sample(embed, grid, mode="bilinear", padding_mode="border", align_corners=False)[0, :, :, 0].t()


# --------------------------------------------------------------------------------------------
# per-frame driver steps
# --------------------------------------------------------------------------------------------
def sot_init(P, cfg, img0: Tensor, box_xyxy: Tensor):
    """unicorn_sot.py:39-55."""
    _, d_pre = forward_backbone(P, cfg, img0)
    H, W = img0.shape[-2:]
    return {"dict_pre": d_pre, "lbs_pre": label_map_s8(box_xyxy, H, W)}


def sot_step(P, cfg, state, img: Tensor, half_corr: bool = False):
    """unicorn_sot.py:78-109 up to (not including) NMS: returns dict of every stage boundary."""
    fpn, d_cur = forward_backbone(P, cfg, img)
    f_pre, f_cur = forward_interaction(P, state["dict_pre"], d_cur)
    e_pre, e_cur = forward_upsample(P, f_pre), forward_upsample(P, f_cur)
    dh, dw = d_cur["h"] * 2, d_cur["w"] * 2
    pred = correlation_propagate(e_pre.flatten(-2)[0], e_cur.flatten(-2)[0], state["lbs_pre"], half_corr)
    coarse = pred.view(1, -1, dh, dw).float()
    pri = prior_pyramid(coarse)
    if cfg.mask:
        out = head_mask_forward(P, cfg, fpn, pri, "sot")
    else:
        out = head_forward(P, cfg, fpn, pri, "sot")
    return {"fpn": fpn, "seq": d_cur, "feat_pre": f_pre, "feat_cur": f_cur, "embed_pre": e_pre,
            "embed_cur": e_cur, "coarse": coarse, "head": out}


def mot_whole(P, cfg, img: Tensor):
    """unicorn.py:133-139 (mode='whole'): zero priors, head(mode='mot')."""
    fpn, d = forward_backbone(P, cfg, img)
    bs, _, H, W = img.shape
    pri = tuple(torch.zeros(bs, 1, H // s, W // s) for s in (8, 16, 32))
    out = head_mask_forward(P, cfg, fpn, pri, "mot") if cfg.mask else head_forward(P, cfg, fpn, pri, "mot")
    return out, d, fpn


def sot_pick_box(det: Optional[Tensor], H: int, W: int, r: float = 1.0, max_inst: int = 3):
    """unicorn_sot.py:62-76: clamp, keep <=3, take index 0, /r, xyxy->xywh, int truncation."""
    if det is None:
        return None
    det = det.clone()
    det[:, 0:4:2] = det[:, 0:4:2].clamp(min=0, max=W)
    det[:, 1:4:2] = det[:, 1:4:2].clamp(min=0, max=H)
    b = det[:max_inst, :4].numpy() / r
    b = b.copy()
    b[:, 2] -= b[:, 0]
    b[:, 3] -= b[:, 1]
    return [int(v) for v in b[0]]


# --------------------------------------------------------------------------------------------
# VOS driver step (external/lib/test/tracker/unicorn_vos.py)
# --------------------------------------------------------------------------------------------
def vos_init(P, cfg, img0: Tensor, boxes_xyxy: dict):
    """unicorn_vos.py:43-68: backbone of the reference frame once, one stride-8 label map per object."""
    _, d_pre = forward_backbone(P, cfg, img0)
    H, W = img0.shape[-2:]
    return {"dict_pre": d_pre, "lbs": {k: label_map_s8(b, H, W) for k, b in boxes_xyxy.items()}}


def vos_group_results(P, cfg, fpn, d_cur, d_pre, lbs: dict, obj_ids, H: int, W: int, conf_thre: float = 0.001,
                      nms_thre: float = 0.65):
    """unicorn_vos.py:157-200 (get_det_results) + the per-object selection of :123-155 for ONE reference group: interaction /
    upsample / one correlation, then PER OBJECT: propagate its label map, head(mode="sot"), postprocess_inst, keep the best
    instance.  Returns {obj_id: (det row (7,) | None, mask (H, W) probabilities at network resolution | None)}."""
    f_pre, f_cur = forward_interaction(P, d_pre, d_cur)
    e_pre, e_cur = forward_upsample(P, f_pre), forward_upsample(P, f_cur)
    dh, dw = d_cur["h"] * 2, d_cur["w"] * 2
    out = {}
    for obj_id in obj_ids:
        pred = correlation_propagate(e_pre.flatten(-2)[0], e_cur.flatten(-2)[0], lbs[obj_id])
        coarse = pred.view(1, -1, dh, dw).float()
        head_out = head_mask_forward(P, cfg, fpn, prior_pyramid(coarse), "sot")
        det, masks = postprocess_inst(cfg, head_out, 1, conf_thre, nms_thre)
        if det is None:
            out[obj_id] = (None, None)
            continue
        det = det.clone()
        det[:, 0:4:2] = det[:, 0:4:2].clamp(min=0, max=W)
        det[:, 1:4:2] = det[:, 1:4:2].clamp(min=0, max=H)
        out[obj_id] = (det[0], masks[0, 0])
    return out


def vos_step(P, cfg, state, img: Tensor, conf_thre: float = 0.001, nms_thre: float = 0.65, max_inst: int = 1):
    """one frame for the objects of the first frame (single reference group): see vos_group_results"""
    fpn, d_cur = forward_backbone(P, cfg, img)
    H, W = img.shape[-2:]
    return vos_group_results(P, cfg, fpn, d_cur, state["dict_pre"], state["lbs"], list(state["lbs"].keys()), H, W, conf_thre, nms_thre)


def vos_track_init(P, cfg, img0: Tensor, boxes_xyxy: dict, out_hw, r: float = 1.0):
    """unicorn_vos.py:43-69: boxes are given on the ORIGINAL image (xyxy), the network sees them scaled by r."""
    _, d_pre = forward_backbone(P, cfg, img0)
    Hn, Wn = img0.shape[-2:]
    return {"groups": [(d_pre, list(boxes_xyxy.keys()))], "lbs": {k: label_map_s8(b * r, Hn, Wn) for k, b in boxes_xyxy.items()},
            "H": int(out_hw[0]), "W": int(out_hw[1])}


def vos_track_frame(P, cfg, state, img: Tensor, info: Optional[dict] = None, r: float = 1.0):
    """unicorn_vos.py:71-121 (track) incl. the reference groups of objects that appear mid-sequence (:79-98): every group =
    (backbone dict of the frame where its objects were introduced, their ids).  info (optional): {"init_object_ids": [...],
    "init_bbox": {id: xyxy on the original image}, "init_mask": (H, W) integer id map}.  Returns the (H, W) uint8 id map."""
    import numpy as np
    info = info or {}
    H, W = state["H"], state["W"]
    Hn, Wn = img.shape[-2:]
    fpn, d_cur = forward_backbone(P, cfg, img)
    final = {}
    for d_pre, ids in state["groups"]:                                              # :80-85
        res = vos_group_results(P, cfg, fpn, d_cur, d_pre, state["lbs"], ids, Hn, Wn)
        for k in ids:                                                               # get_mask_results :123-155
            m = res[k][1]
            if m is None:
                final[k] = np.zeros((H, W), dtype=np.uint8)
                continue
            up = F.interpolate(m[None, None], scale_factor=1 / r, mode="bilinear", align_corners=False)[:, 0, :H, :W]
            full = np.zeros((H, W), dtype=np.float32)
            full[:up.shape[1], :up.shape[2]] = up[0].numpy()
            final[k] = full
    cur_ids = [k for _, ids in state["groups"] for k in ids]
    if "init_object_ids" in info:                                                   # :87-98
        state["groups"].append((d_cur, list(info["init_object_ids"])))
        for k in info["init_object_ids"]:
            state["lbs"][k] = label_map_s8(info["init_bbox"][k] * r, Hn, Wn)
            final[k] = (np.asarray(info["init_mask"]) == int(k))
        cur_ids = cur_ids + list(info["init_object_ids"])
    return vos_merge({k: final[k] for k in cur_ids}, H, W)                           # :99-120


def vos_merge(prob: dict, H: int, W: int):
    """unicorn_vos.py:105-121 soft aggregation: background = prod(1 - p_k), argmax over [background, objects] -> (H,W) uint8
    object-id map.  prob: {obj_id (str/int): (H, W) float array}."""
    import numpy as np
    ids = [int(k) for k in prob]
    merge = np.zeros((H, W, max(ids) + 1))
    for k, p in prob.items():
        merge[:, :, int(k)] = p
    merge[:, :, 0] = np.prod(1 - np.stack(list(prob.values()), axis=-1), axis=-1)
    lab = np.argmax(merge, axis=-1)
    final = np.zeros((H, W), dtype=np.uint8)
    for k in ids:
        final[lab == k] = k
    return final
