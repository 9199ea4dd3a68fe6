"""Checkpoint tooling (row N4 of SURVEY.md §8f) with the reference's function names:
  * load_ckpt               unicorn/utils/checkpoint.py:11-33  (load what matches by name AND shape, warn about the rest)
  * remap_coco_pretrain     unicorn/exp/unicorn_track.py:157-190 (COCO detector checkpoint -> unified track head)
  * state_spec              the learnable tensors of exp.get_model() (name -> shape) for a model configuration; held to the
                            specs dumped from the REAL reference models (tests/golden/state_spec_*.json)
Released checkpoints are `{"model": state_dict, ...}` files (tools/track.py:186-188): load_checkpoint_file unwraps them.
"""
import logging
from collections import OrderedDict

import torch

logger = logging.getLogger("unicorn_amd")


def _block(s, p, c):              # convnext.py:16-39 Block
    s[p + "dwconv.weight"] = (c, 1, 7, 7); s[p + "dwconv.bias"] = (c,)
    s[p + "norm.weight"] = (c,); s[p + "norm.bias"] = (c,)
    s[p + "pwconv1.weight"] = (4 * c, c); s[p + "pwconv1.bias"] = (4 * c,)
    s[p + "pwconv2.weight"] = (c, 4 * c); s[p + "pwconv2.bias"] = (c,)
    s[p + "gamma"] = (c,)


def _base_conv(s, p, cin, cout, k):    # network_blocks.py:29-51 BaseConv with GroupNorm (exp/unicorn_track.py:150-155)
    s[p + "conv.weight"] = (cout, cin, k, k)
    s[p + "bn.weight"] = (cout,); s[p + "bn.bias"] = (cout,)


def _csp(s, p, cin, cout, n=3):        # network_blocks.py:147-185 CSPLayer
    h = cout // 2
    _base_conv(s, p + "conv1.", cin, h, 1)
    _base_conv(s, p + "conv2.", cin, h, 1)
    _base_conv(s, p + "conv3.", 2 * h, cout, 1)
    for i in range(n):
        _base_conv(s, p + "m.%d.conv1." % i, h, h, 1)
        _base_conv(s, p + "m.%d.conv2." % i, h, h, 3)


def state_spec(name_or_cfg):
    """OrderedDict name -> shape tuple of every learnable tensor of the reference model for this configuration"""
    from ..models.unicorn import MODEL_CONFIGS
    cfg = dict(MODEL_CONFIGS[name_or_cfg]) if isinstance(name_or_cfg, str) else dict(name_or_cfg)
    d, depths, nc, mask = cfg["dims"], cfg["depths"], cfg["num_classes"], cfg["mask"]
    n_att, embed, up_rate = cfg.get("n_layer_att", 3), cfg.get("embed_dim", 128), cfg.get("up_rate", 4)
    s = OrderedDict()
    bb = "backbone.backbone."
    s[bb + "downsample_layers.0.0.weight"] = (d[0], 3, 4, 4); s[bb + "downsample_layers.0.0.bias"] = (d[0],)
    s[bb + "downsample_layers.0.1.weight"] = (d[0],); s[bb + "downsample_layers.0.1.bias"] = (d[0],)
    for i in range(1, 4):
        s[bb + "downsample_layers.%d.0.weight" % i] = (d[i - 1],); s[bb + "downsample_layers.%d.0.bias" % i] = (d[i - 1],)
        s[bb + "downsample_layers.%d.1.weight" % i] = (d[i], d[i - 1], 2, 2); s[bb + "downsample_layers.%d.1.bias" % i] = (d[i],)
    for i in range(4):
        for j in range(depths[i]):
            _block(s, bb + "stages.%d.%d." % (i, j), d[i])
    for i in (1, 2, 3):
        s[bb + "norm%d.weight" % i] = (d[i],); s[bb + "norm%d.bias" % i] = (d[i],)
    c0, c1, c2 = d[1:]
    b = "backbone."
    _base_conv(s, b + "lateral_conv0.", c2, c1, 1); _csp(s, b + "C3_p4.", 2 * c1, c1)
    _base_conv(s, b + "reduce_conv1.", c1, c0, 1); _csp(s, b + "C3_p3.", 2 * c0, c0)
    _base_conv(s, b + "bu_conv2.", c0, c0, 3); _csp(s, b + "C3_n3.", 2 * c0, c1)
    _base_conv(s, b + "bu_conv1.", c1, c1, 3); _csp(s, b + "C3_n4.", 2 * c1, c2)
    h = "head."
    for k in range(3):
        s[h + "beta_%d" % k] = (256, 1, 1)
    for tower in ("cls_convs", "reg_convs"):
        for k in range(3):
            for i in range(4):
                _base_conv(s, h + "%s.%d.%d." % (tower, k, i), 256, 256, 3)
    for name, n in (("cls_preds", nc), ("reg_preds", 4), ("obj_preds", 1), ("cls_preds_sot", 1), ("obj_preds_sot", 1),
                    ("reg_preds_sot", 4)):
        for k in range(3):
            s[h + "%s.%d.weight" % (name, k)] = (n, 256, 1, 1); s[h + "%s.%d.bias" % (name, k)] = (n,)
    if mask:
        mb = h + "mask_branch."
        for k, c in enumerate((c0, c1, c2)):
            s[mb + "refine.%d.0.weight" % k] = (128, c, 3, 3)
            s[mb + "refine.%d.1.weight" % k] = (128,); s[mb + "refine.%d.1.bias" % k] = (128,)
        for i in range(4):
            s[mb + "tower.%d.0.weight" % i] = (128, 128, 3, 3)
            s[mb + "tower.%d.1.weight" % i] = (128,); s[mb + "tower.%d.1.bias" % i] = (128,)
        s[mb + "tower.4.weight"] = (8, 128, 1, 1); s[mb + "tower.4.bias"] = (8,)
        s[mb + "up_mask_layer.0.weight"] = (128, 128, 3, 3); s[mb + "up_mask_layer.0.bias"] = (128,)
        s[mb + "up_mask_layer.2.weight"] = (9 * up_rate ** 2, 128, 1, 1); s[mb + "up_mask_layer.2.bias"] = (9 * up_rate ** 2,)
        for k in range(3):
            s[h + "controllers.%d.weight" % k] = (169, 256, 3, 3); s[h + "controllers.%d.bias" % k] = (169,)
    for k, c in enumerate((c0, c1, c2)):
        _base_conv(s, h + "stems.%d." % k, c, 256, 1)
    for k in range(3):
        for n in range(n_att):
            _block(s, h + "att_layers.%d.%d." % (k, n), 256)
    s["bottleneck.0.weight"] = (256, c1, 1, 1); s["bottleneck.0.bias"] = (256,)
    s["bottleneck.1.weight"] = (256,); s["bottleneck.1.bias"] = (256,)
    s["upsample_layer.1.weight"] = (256, 64, 3, 3); s["upsample_layer.1.bias"] = (256,)
    s["upsample_layer.3.weight"] = (embed, 256, 3, 3); s["upsample_layer.3.bias"] = (embed,)
    s["pos_emb.row_embed.weight"] = (40, 128); s["pos_emb.col_embed.weight"] = (40, 128)
    s["transformer.level_embed"] = (2, 256)
    e = "transformer.encoder.layers.0."
    for nm, shp in (("self_attn.sampling_offsets", (128, 256)), ("self_attn.attention_weights", (64, 256)),
                    ("self_attn.value_proj", (256, 256)), ("self_attn.output_proj", (256, 256)), ("linear1", (1024, 256)),
                    ("linear2", (256, 1024))):
        s[e + nm + ".weight"] = shp; s[e + nm + ".bias"] = (shp[0],)
    for nm in ("norm1", "norm2"):
        s[e + nm + ".weight"] = (256,); s[e + nm + ".bias"] = (256,)
    return s


def filter_ckpt(spec, ckpt):
    """the selection rule of load_ckpt (checkpoint.py:12-30): -> (load_dict, missing names, shape-mismatched names)"""
    load_dict, missing, mismatched = OrderedDict(), [], []
    for key, shape in spec.items():
        if key not in ckpt:
            missing.append(key)
            continue
        if tuple(ckpt[key].shape) != tuple(shape):
            mismatched.append(key)
            continue
        load_dict[key] = ckpt[key]
    return load_dict, missing, mismatched


def load_ckpt(model, ckpt):
    """unicorn/utils/checkpoint.py:11-33"""
    if hasattr(model, "dims"):      # unicorn_amd.models.Unicorn
        spec = state_spec(dict(dims=model.dims, depths=model.depths, num_classes=model.num_classes, mask=model.mask,
                               n_layer_att=model.n_layer_att, embed_dim=model.embed_dim, up_rate=model.up_rate))
    else:
        spec = state_spec(model.name)
    load_dict, missing, mismatched = filter_ckpt(spec, ckpt)
    for k in missing:
        logger.warning("%s is not in the ckpt. Please double check and see if this is desired.", k)
    for k in mismatched:
        logger.warning("Shape of %s in checkpoint is %s, while shape of %s in model is %s.", k, tuple(ckpt[k].shape), k, spec[k])
    model.load_state_dict(load_dict, strict=False)
    return model


def load_checkpoint_file(model, path, strict=False):
    """tools/track.py:186-188: ckpt = torch.load(file, map_location="cpu"); model.load_state_dict(ckpt["model"], strict=False)"""
    try:                                                            # tensors-only unpickling first: the file is user-supplied
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
    except Exception:                                               # noqa: BLE001 -- pickled python objects: the reference's own call
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
    sd = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt else ckpt
    return model.load_state_dict(sd, strict=strict)


_COCO_TO_BDD = [0, 0, 2, 7, 5, 6, 3, 1]        # exp/unicorn_track.py:173
_CLS_KEYS = {"head.cls_preds.%d.%s" % (k, t) for k in range(3) for t in ("weight", "bias")}


def remap_coco_pretrain(state_dict, num_classes, unshared_obj=True, unshared_reg=True):
    """exp/unicorn_track.py:163-186: the 80-class COCO cls_preds are reduced to the tracker's classes (8: the listed COCO
    indices, 1: class 0), obj/reg preds are duplicated into the *_sot heads, everything else is copied."""
    out = OrderedDict()
    for k, v in state_dict.items():
        if not k.startswith("head."):
            out[k] = v
        elif k in _CLS_KEYS:
            if num_classes == 8:
                out[k] = v[_COCO_TO_BDD]
            elif num_classes == 1:
                out[k] = v[0:1]
            else:
                raise ValueError("Invalid num_classes")
        elif unshared_obj and k.startswith("head.obj_preds."):
            out[k] = v
            out[k.replace("head.obj_preds.", "head.obj_preds_sot.")] = v
        elif unshared_reg and k.startswith("head.reg_preds."):
            out[k] = v
            out[k.replace("head.reg_preds.", "head.reg_preds_sot.")] = v
        else:
            out[k] = v
    return out


def export_flat(state_dict, model_or_cfg, path, precision="f16x2"):
    """Flat weights file for hosts without Python / torch (include/unicorn_hip.h: uni_weights_file_cfg / uni_ctx_load_file) -- the deployment artefact of this
    backend, in the role `tools/export_torchscript.py:51-71` has in the reference (a file another runtime loads).  Layout: b"UNIW1\0\0\0", the uni_model_cfg as 15
    int32, int32 tensor count, then per learnable tensor of the experiment's spec (reference names, reference layout): int32 name length, name, int32 ndim,
    int64 shape[ndim], fp32 data.  `model_or_cfg`: an experiment name ("unicorn_track_large"), a config dict or a `unicorn_amd.models.Unicorn` instance.
    Only tensors of the spec are written (buffers / foreign keys are skipped like load_state_dict(strict=False) ignores them); a spec tensor missing from the
    state dict raises.  -> number of tensors written."""
    import struct
    import numpy as np
    from ..models.unicorn import MODEL_CONFIGS, PRECISIONS
    if hasattr(model_or_cfg, "dims") and hasattr(model_or_cfg, "depths") and not isinstance(model_or_cfg, dict):
        m = model_or_cfg
        cfg = dict(dims=m.dims, depths=m.depths, num_classes=m.num_classes, mask=m.mask, n_layer_att=m.n_layer_att, embed_dim=m.embed_dim, d_rate=m.d_rate)
    else:
        cfg = dict(MODEL_CONFIGS[model_or_cfg]) if isinstance(model_or_cfg, str) else dict(model_or_cfg)
    d_rate = int(cfg.get("d_rate", 2))
    full = dict(dims=tuple(cfg["dims"]), depths=tuple(cfg["depths"]), num_classes=int(cfg["num_classes"]), mask=bool(cfg["mask"]),
                n_layer_att=int(cfg.get("n_layer_att", 3)), embed_dim=int(cfg.get("embed_dim", 128)), up_rate=8 // d_rate)
    spec = state_spec(full)
    missing = [k for k in spec if k not in state_dict]
    if missing:
        raise KeyError("export_flat: %d tensors of the experiment are not in the state dict, e.g. %s" % (len(missing), missing[:4]))
    head = struct.pack("<15i", *full["dims"], *full["depths"], full["num_classes"], int(full["mask"]), full["n_layer_att"], full["embed_dim"], full["up_rate"], d_rate,
                       PRECISIONS[precision])
    with open(path, "wb") as f:
        f.write(b"UNIW1\0\0\0")
        f.write(head)
        f.write(struct.pack("<i", len(spec)))
        for k, shp in spec.items():
            a = np.ascontiguousarray(state_dict[k].detach().float().cpu().numpy(), dtype="<f4")
            if tuple(a.shape) != tuple(shp):
                raise ValueError("export_flat: %s has shape %s, the experiment expects %s" % (k, tuple(a.shape), tuple(shp)))
            kb = k.encode()
            f.write(struct.pack("<i", len(kb)))
            f.write(kb)
            f.write(struct.pack("<i", a.ndim))
            f.write(struct.pack("<%dq" % a.ndim, *a.shape))
            f.write(a.tobytes())
    return len(spec)

