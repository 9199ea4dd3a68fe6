"""Deterministic synthetic weights and clips (no checkpoints/datasets exist offline).

TEST INFRASTRUCTURE (also used by bench.py to fill the model with weights of the right shape).
`param_spec(cfg)` restates the reference state-dict layout (names + shapes; checked against the real
reference models through tests/golden/state_spec_*.json, dumped by tests/golden/make_golden.py);
`synth_state_dict(cfg)` fills every tensor from a per-name seeded generator so the build container
(real reference) and the GPU box (oracle + HIP path) see bit-identical weights without shipping them.
"""
import math
import zlib
from collections import OrderedDict

import torch

from unicorn_oracle import ModelCfg


def _block(spec, p, C):
    spec[p + "gamma"] = (C,)
    spec[p + "dwconv.weight"] = (C, 1, 7, 7)
    spec[p + "dwconv.bias"] = (C,)
    spec[p + "norm.weight"] = (C,)
    spec[p + "norm.bias"] = (C,)
    spec[p + "pwconv1.weight"] = (4 * C, C)
    spec[p + "pwconv1.bias"] = (4 * C,)
    spec[p + "pwconv2.weight"] = (C, 4 * C)
    spec[p + "pwconv2.bias"] = (C,)


def _base_conv(spec, p, cin, cout, k):
    spec[p + "conv.weight"] = (cout, cin, k, k)
    spec[p + "bn.weight"] = (cout,)
    spec[p + "bn.bias"] = (cout,)


def _csp(spec, p, cin, cout, n=3):
    h = cout // 2
    _base_conv(spec, p + "conv1.", cin, h, 1)
    _base_conv(spec, p + "conv2.", cin, h, 1)
    _base_conv(spec, p + "conv3.", 2 * h, cout, 1)
    for i in range(n):
        _base_conv(spec, p + "m.%d.conv1." % i, h, h, 1)
        _base_conv(spec, p + "m.%d.conv2." % i, h, h, 3)


def param_spec(cfg: ModelCfg) -> "OrderedDict[str, tuple]":
    """Names/shapes of every learnable tensor of exp.get_model() (buffers excluded)."""
    s = OrderedDict()
    d = cfg.dims
    bb = "backbone.backbone."
    s[bb + "downsample_layers.0.0.weight"] = (d[0], 3, 4, 4)
    s[bb + "downsample_layers.0.0.bias"] = (d[0],)
    s[bb + "downsample_layers.0.1.weight"] = (d[0],)
    s[bb + "downsample_layers.0.1.bias"] = (d[0],)
    for i in range(1, 4):
        s[bb + "downsample_layers.%d.0.weight" % i] = (d[i - 1],)
        s[bb + "downsample_layers.%d.0.bias" % i] = (d[i - 1],)
        s[bb + "downsample_layers.%d.1.weight" % i] = (d[i], d[i - 1], 2, 2)
        s[bb + "downsample_layers.%d.1.bias" % i] = (d[i],)
    for i in range(4):
        for j in range(cfg.depths[i]):
            _block(s, bb + "stages.%d.%d." % (i, j), d[i])
    for i in (1, 2, 3):
        s[bb + "norm%d.weight" % i] = (d[i],)
        s[bb + "norm%d.bias" % i] = (d[i],)
    c0, c1, c2 = cfg.in_channels
    b = "backbone."
    _base_conv(s, b + "lateral_conv0.", c2, c1, 1)
    _csp(s, b + "C3_p4.", 2 * c1, c1)
    _base_conv(s, b + "reduce_conv1.", c1, c0, 1)
    _csp(s, b + "C3_p3.", 2 * c0, c0)
    _base_conv(s, b + "bu_conv2.", c0, c0, 3)
    _csp(s, b + "C3_n3.", 2 * c0, c1)
    _base_conv(s, b + "bu_conv1.", c1, c1, 3)
    _csp(s, b + "C3_n4.", 2 * c1, c2)
    h = "head."
    for k in range(3):
        s[h + "beta_%d" % k] = (256, 1, 1)
    for k in range(3):
        for i in range(4):
            _base_conv(s, h + "cls_convs.%d.%d." % (k, i), 256, 256, 3)
    for k in range(3):
        for i in range(4):
            _base_conv(s, h + "reg_convs.%d.%d." % (k, i), 256, 256, 3)
    for name, n in (("cls_preds", cfg.num_classes), ("reg_preds", 4), ("obj_preds", 1),
                    ("cls_preds_sot", 1), ("obj_preds_sot", 1), ("reg_preds_sot", 4)):
        for k in range(3):
            s[h + "%s.%d.weight" % (name, k)] = (n, 256, 1, 1)
            s[h + "%s.%d.bias" % (name, k)] = (n,)
    if cfg.mask:
        mb = h + "mask_branch."
        for k, c in enumerate(cfg.in_channels):
            s[mb + "refine.%d.0.weight" % k] = (128, c, 3, 3)
            s[mb + "refine.%d.1.weight" % k] = (128,)
            s[mb + "refine.%d.1.bias" % k] = (128,)
        for i in range(4):
            s[mb + "tower.%d.0.weight" % i] = (128, 128, 3, 3)
            s[mb + "tower.%d.1.weight" % i] = (128,)
            s[mb + "tower.%d.1.bias" % i] = (128,)
        s[mb + "tower.4.weight"] = (8, 128, 1, 1)
        s[mb + "tower.4.bias"] = (8,)
        s[mb + "up_mask_layer.0.weight"] = (128, 128, 3, 3)
        s[mb + "up_mask_layer.0.bias"] = (128,)
        s[mb + "up_mask_layer.2.weight"] = (9 * cfg.up_rate ** 2, 128, 1, 1)
        s[mb + "up_mask_layer.2.bias"] = (9 * cfg.up_rate ** 2,)
        for k in range(3):
            s[h + "controllers.%d.weight" % k] = (169, 256, 3, 3)
            s[h + "controllers.%d.bias" % k] = (169,)
    for k, c in enumerate(cfg.in_channels):
        _base_conv(s, h + "stems.%d." % k, c, 256, 1)
    for k in range(3):
        for n in range(cfg.n_layer_att):
            _block(s, h + "att_layers.%d.%d." % (k, n), 256)
    s["bottleneck.0.weight"] = (256, c1, 1, 1)
    s["bottleneck.0.bias"] = (256,)
    s["bottleneck.1.weight"] = (256,)
    s["bottleneck.1.bias"] = (256,)
    s["upsample_layer.1.weight"] = (256, 64, 3, 3)
    s["upsample_layer.1.bias"] = (256,)
    s["upsample_layer.3.weight"] = (cfg.embed_dim, 256, 3, 3)
    s["upsample_layer.3.bias"] = (cfg.embed_dim,)
    s["pos_emb.row_embed.weight"] = (40, 128)
    s["pos_emb.col_embed.weight"] = (40, 128)
    s["transformer.level_embed"] = (2, 256)
    e = "transformer.encoder.layers.0."
    for nm, shp in (("self_attn.sampling_offsets", (128, 256)), ("self_attn.attention_weights", (64, 256)),
                    ("self_attn.value_proj", (256, 256)), ("self_attn.output_proj", (256, 256))):
        s[e + nm + ".weight"] = shp
        s[e + nm + ".bias"] = (shp[0],)
    s[e + "norm1.weight"] = (256,)
    s[e + "norm1.bias"] = (256,)
    s[e + "linear1.weight"] = (1024, 256)
    s[e + "linear1.bias"] = (1024,)
    s[e + "linear2.weight"] = (256, 1024)
    s[e + "linear2.bias"] = (256,)
    s[e + "norm2.weight"] = (256,)
    s[e + "norm2.bias"] = (256,)
    return s


def _gen(name, salt=0):
    g = torch.Generator()
    g.manual_seed((zlib.crc32(name.encode()) + 7919 * salt) & 0x7FFFFFFF)
    return g


def synth_param(name: str, shape, seed: int = 0) -> torch.Tensor:
    """One synthetic parameter. Scales are chosen so activations stay O(1) through ~100 layers, every
    bias/affine path is exercised (nothing is exactly 0 or 1) and a few % of anchors pass the
    reference's score thresholds so NMS / mask code runs."""
    g = _gen(name, seed)
    n = lambda std=1.0: torch.randn(shape, generator=g) * std
    if name.startswith("pos_emb."):
        return torch.rand(shape, generator=g)                     # position_encoding.py:21-23
    if name == "transformer.level_embed":
        return n()
    if name.endswith("sampling_offsets.bias"):                   # ms_deform_attn.py:62-70 grid init + jitter
        th = torch.arange(8, dtype=torch.float32) * (2.0 * math.pi / 8)
        gi = torch.stack([th.cos(), th.sin()], -1)
        gi = (gi / gi.abs().max(-1, keepdim=True)[0]).view(8, 1, 1, 2).repeat(1, 2, 4, 1)
        for i in range(4):
            gi[:, :, i, :] *= i + 1
        return gi.reshape(-1) + n(0.1)
    if name.endswith("sampling_offsets.weight") or name.endswith("attention_weights.weight"):
        return n(0.02)
    if name.startswith("head.beta_") or name.endswith("gamma"):
        return 1.0 + n(0.1)
    if any(name.startswith("head.%s" % k) for k in ("cls_preds", "obj_preds")) and name.endswith("bias"):
        return -4.5 + n(0.1)
    if len(shape) == 1:
        if name.endswith("weight"):
            return 1.0 + n(0.1)                                  # norm scales
        return n(0.05)                                           # biases / norm shifts
    fan_in = 1
    for v in shape[1:]:
        fan_in *= v
    gain = 0.5 if name.endswith("pwconv2.weight") else 1.0
    return n(gain / math.sqrt(fan_in))


def synth_state_dict(cfg: ModelCfg, seed: int = 0):
    return OrderedDict((k, synth_param(k, shp, seed)) for k, shp in param_spec(cfg).items())


def synth_clip(H: int, W: int, n_frames: int = 2, seed: int = 1):
    """frame0 = rand*255; frame t = frame0 rolled by (3t,5t) px + U(0,8) noise.
    Returns list of (1,3,H,W) fp32 tensors (BGR 0-255 as the reference expects, normalize=False) and
    the init box xyxy = (W/4, H/4, W/2, H/2)."""
    g = torch.Generator()
    g.manual_seed(seed)
    # smooth-ish content: low-res noise upsampled + fine noise, so correspondence is meaningful
    base = torch.nn.functional.interpolate(torch.rand(1, 3, H // 16 + 1, W // 16 + 1, generator=g),
                                           size=(H, W), mode="bilinear", align_corners=False)
    f0 = (0.7 * base + 0.3 * torch.rand(1, 3, H, W, generator=g)) * 255.0
    frames = [f0]
    for t in range(1, n_frames):
        f = torch.roll(f0, shifts=(3 * t, 5 * t), dims=(2, 3)) + torch.rand(1, 3, H, W, generator=g) * 8.0
        frames.append(f.clamp(0, 255))
    box = torch.tensor([W / 4.0, H / 4.0, W / 2.0, H / 2.0])
    return frames, box
