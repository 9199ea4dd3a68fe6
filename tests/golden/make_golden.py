"""Generate golden vectors from the REAL reference (run in the build container only).

    python tests/golden/make_golden.py

Imports /root/reference through oracle/ref_bootstrap.py, loads the synthetic weights of
oracle/synth.py into the reference nn.Modules (strict), runs the reference's own forward modes and
its post-processing on the synthetic clip (config 0 of BASELINE.json: tiny, 2-frame 320x320, CPU),
and stores every stage boundary.  Large tensors are stored as a fixed strided sample plus moments.
The driver tensor logic (unicorn_sot.py:82-108) is executed here with the reference model's own
modes; only `.cuda()`/fp16 casts are dropped (CPU, fp32 correlation = the parity definition).
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import ref_bootstrap as rb  # noqa: E402
import synth  # noqa: E402
import unicorn_oracle as uo  # noqa: E402

MAX_FULL = 1 << 15          # samples per tensor of the 320-class goldens; the 800x1280 files store `max_full` themselves (key "__max_full")
_max_full = MAX_FULL


def pack(out, name, t):
    t = t.detach().float().contiguous()
    a = t.numpy().reshape(-1)
    out[name + "__shape"] = np.array(t.shape, dtype=np.int64)
    out[name + "__stats"] = np.array([a.mean(), np.abs(a).mean(), a.min(), a.max()], dtype=np.float64)
    if a.size <= _max_full:
        out[name] = a.astype(np.float32)
    else:
        idx = np.linspace(0, a.size - 1, _max_full).astype(np.int64)
        out[name] = a[idx].astype(np.float32)


def load_synth(model, cfg):
    sd = synth.synth_state_dict(cfg)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    # only buffers may be missing (mask_head.sizes_of_interest / _iter)
    assert all("mask_head" in m for m in missing), missing
    assert not unexpected, unexpected
    return sd


def run_sot(exp_name, H, W, max_full=MAX_FULL, top_rows=0):
    """max_full: samples kept per tensor (the headline-size files keep 6144 so that one file stays ~0.5 MB); top_rows > 0: also store the
    `top_rows` best-scoring raw head rows (index + row), since a strided sample of a (1, 21000, 6) output cannot be turned into boxes."""
    global _max_full
    _max_full = max_full
    cfg = uo.CONFIGS[exp_name]
    model, exp = rb.build_reference_model(exp_name)
    load_synth(model, cfg)
    frames, box = synth.synth_clip(H, W, 2, seed=1)
    out = {"__max_full": np.array([max_full], dtype=np.int64)}
    with torch.no_grad():
        # --- initialize (unicorn_sot.py:39-55) ---
        _, d_pre = model(imgs=frames[0], mode="backbone")
        from unicorn.utils.boxes import postprocess, postprocess_inst  # reference post-processing
        lab = torch.zeros((1, 1, H, W))
        x1, y1, x2, y2 = torch.round(box).int().tolist()
        lab[0, 0, max(0, min(y1, H)):max(0, min(y2, H)), max(0, min(x1, W)):max(0, min(x2, W))] = 1.0
        lbs_pre = F.interpolate(lab, scale_factor=1 / 8, mode="bilinear", align_corners=False)[0].flatten(-2)
        pack(out, "lbs_pre", lbs_pre)
        # --- track (unicorn_sot.py:78-108) ---
        fpn, d_cur = model(imgs=frames[1], mode="backbone")
        for i, f in enumerate(fpn):
            pack(out, "fpn%d" % i, f)
        pack(out, "seq_feat", d_cur["feat"])
        pack(out, "seq_pos", d_cur["pos"])
        f_pre, f_cur = model(seq_dict0=d_pre, seq_dict1=d_cur, mode="interaction")
        pack(out, "feat_pre", f_pre)
        pack(out, "feat_cur", f_cur)
        e_pre = model(feat=f_pre, mode="upsample")
        e_cur = model(feat=f_cur, mode="upsample")
        pack(out, "embed_pre", e_pre)
        pack(out, "embed_cur", e_cur)
        keys, q = e_pre.flatten(-2).squeeze(), e_cur.flatten(-2).squeeze()
        simi = torch.mm(keys.transpose(1, 0), q)
        trans = torch.softmax(simi, dim=0)
        cur_pred = lbs_pre @ trans
        dh, dw = d_cur["h"] * 2, d_cur["w"] * 2
        coarse = cur_pred.view(1, -1, dh, dw).float()
        pack(out, "coarse", coarse)
        pri = (coarse, F.interpolate(coarse, scale_factor=1 / 2, mode="bilinear", align_corners=False),
               F.interpolate(coarse, scale_factor=1 / 4, mode="bilinear", align_corners=False))
        pack(out, "prior16", pri[1])
        pack(out, "prior32", pri[2])
        head_out = model.head(fpn, pri, mode="sot")
        if top_rows:
            ho = head_out[0] if cfg.mask else head_out
            top = torch.argsort(ho[0, :, 4] * ho[0, :, 5], descending=True)[:top_rows]
            out["head_top_idx"] = top.numpy().astype(np.int64)
            out["head_top_rows"] = ho[0, top].numpy().astype(np.float32)
            if cfg.mask:
                out["dyn_top_rows"] = head_out[2][0, top[:64]].numpy().astype(np.float32)
        if cfg.mask:
            names = ["head_out", "locations", "dyn_params", "fpn_levels", "mask_feats", "up_masks"]
            for n, t in zip(names, head_out):
                pack(out, n, t)
            dets, masks = postprocess_inst(head_out[0].clone(), head_out[1], head_out[2], head_out[3], head_out[4],
                                           model.head.mask_head, 1, 0.001, 0.65, d_rate=cfg.d_rate,
                                           up_masks=head_out[5])
            det, msk = dets[0], masks[0]
            det, msk = (det[:8], msk[:8]) if not top_rows else (det[:4], msk[:4])
            pack(out, "det_sot", det)
            out["mask_sot_bits"] = np.packbits((msk > 0.5).numpy().astype(np.uint8).reshape(-1))
            pack(out, "mask_sot", msk[:, :, ::8, ::8])
            out["n_det_sot"] = np.array([dets[0].shape[0]])
        else:
            pack(out, "head_out", head_out)
            det = postprocess(head_out.clone(), 1, 0.001, 0.65)[0]
            out["n_det_sot"] = np.array([0 if det is None else det.shape[0]])
            if det is not None:
                pack(out, "det_sot", det[:64])
        # --- MOT entry (unicorn.py:133-139) on frame 1 ---
        whole, _ = model(frames[1])
        who = whole[0] if cfg.mask else whole
        pack(out, "whole_out", who)
        if top_rows:
            topw = torch.argsort(who[0, :, 4] * who[0, :, 5:].max(1)[0], descending=True)[:top_rows]
            out["whole_top_idx"] = topw.numpy().astype(np.int64)
            out["whole_top_rows"] = who[0, topw].numpy().astype(np.float32)
        det = postprocess(who.clone(), cfg.num_classes, 0.0005, 0.65)[0]
        out["n_det_mot"] = np.array([0 if det is None else det.shape[0]])
        if det is not None:
            pack(out, "det_mot", det[:64])
            # instance embeddings at box centres: the reference's own lines (mot_evaluator.py:1024-1034) exec'd verbatim
            import textwrap
            import types
            src = open("/root/reference/unicorn/evaluators/mot_evaluator.py").read().split("\n")[1023:1034]
            assert "cx, cy = (bboxes[:, 0] + bboxes[:, 2])/2/s - 0.5" in src[0] and "track_feats = torch.stack" in src[-1]
            ns = {"torch": torch, "F": F, "bboxes": det[:16, :4].clone(), "s": 8, "embed_cur": e_cur,
                  "self": types.SimpleNamespace(img_size=(H, W))}
            exec("track_feat_list = []\n" + textwrap.dedent("\n".join(src)), ns)
            embs = list(ns["track_feats"])
            pack(out, "inst_embed", torch.stack(embs))
    _max_full = MAX_FULL
    np.savez_compressed(os.path.join(HERE, "%s_%dx%d.npz" % (exp_name, H, W)), **out)
    print(exp_name, {k: v.shape for k, v in out.items() if not k.endswith("__shape") and not k.endswith("__stats")})


def run_vos(exp_name, H, W):
    """get_det_results of the VOS driver (external/lib/test/tracker/unicorn_vos.py:157-200) with the REAL reference model's modes and the
    reference's own postprocess_inst: one interaction / upsample / correlation for the frame pair, then PER OBJECT label-map
    propagation, head(mode="sot"), CondInst masks; the best instance per object (:131-152) is stored.  Only `.cuda()` / fp16 casts are
    dropped (CPU, fp32 correlation = the parity definition)."""
    cfg = uo.CONFIGS[exp_name]
    model, exp = rb.build_reference_model(exp_name)
    load_synth(model, cfg)
    from unicorn.utils.boxes import postprocess_inst
    frames, box = synth.synth_clip(H, W, 2, seed=1)
    boxes = {"1": box, "2": torch.tensor([W * 0.55, H * 0.1, W * 0.9, H * 0.45]), "3": torch.tensor([W * 0.1, H * 0.55, W * 0.4, H * 0.95])}
    out = {}
    with torch.no_grad():
        _, d_pre = model(imgs=frames[0], mode="backbone")
        lbs = {}
        for k, b in boxes.items():                                  # unicorn_vos.py:52-56 (get_label_map + 1/8 bilinear)
            lab = torch.zeros((1, 1, H, W))
            x1, y1, x2, y2 = torch.round(b).int().tolist()
            lab[0, 0, max(0, min(y1, H)):max(0, min(y2, H)), max(0, min(x1, W)):max(0, min(x2, W))] = 1.0
            lbs[k] = F.interpolate(lab, scale_factor=1 / 8, mode="bilinear", align_corners=False)[0].flatten(-2)
        fpn, d_cur = model(imgs=frames[1], mode="backbone")
        f_pre, f_cur = model(seq_dict0=d_pre, seq_dict1=d_cur, mode="interaction")          # :162
        e_pre = model(feat=f_pre, mode="upsample").flatten(-2).squeeze()                        # :164-167
        e_cur = model(feat=f_cur, mode="upsample").flatten(-2).squeeze()
        trans = torch.softmax(torch.mm(e_pre.transpose(1, 0), e_cur), dim=0)                    # :173-174
        dh, dw = d_cur["h"] * 2, d_cur["w"] * 2
        for k in boxes:                                                                         # :178-200
            coarse = (lbs[k] @ trans).view(1, -1, dh, dw).float()
            pri = (coarse, F.interpolate(coarse, scale_factor=1 / 2, mode="bilinear", align_corners=False),
                   F.interpolate(coarse, scale_factor=1 / 4, mode="bilinear", align_corners=False))
            o, loc, dyn, lev, mf, um = model.head(fpn, pri, mode="sot")
            dets, masks = postprocess_inst(o, loc, dyn, lev, mf, model.head.mask_head, 1, 0.001, 0.65, class_agnostic=False,
                                           d_rate=cfg.d_rate, up_masks=um[0:1])
            out["n_det_%s" % k] = np.array([0 if dets[0] is None else dets[0].shape[0]])
            if dets[0] is not None:
                pack(out, "det_%s" % k, dets[0][0])
                out["mask_bits_%s" % k] = np.packbits((masks[0][0, 0] > 0.5).numpy().astype(np.uint8).reshape(-1))
                pack(out, "mask_%s" % k, masks[0][0, 0][::4, ::4])
    np.savez_compressed(os.path.join(HERE, "%s_vos_%dx%d.npz" % (exp_name, H, W)), **out)
    print(exp_name, "vos", {k: v.shape for k, v in out.items() if not k.endswith("__shape") and not k.endswith("__stats")})


def run_msda_known_answer():
    """unicorn/models/ops/test.py:24-50 shapes & seed; answer = the reference's own pure-PyTorch core."""
    rb.boot()
    from unicorn.models.ops.functions.ms_deform_attn_func import ms_deform_attn_core_pytorch
    N, M, D = 1, 2, 2
    Lq, L, P = 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    S = sum([(H * W).item() for H, W in shapes])
    torch.manual_seed(3)
    value = torch.rand(N, S, M, D) * 0.01
    loc = torch.rand(N, Lq, M, L, P, 2)
    attn = torch.rand(N, Lq, M, L, P) + 1e-5
    attn /= attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
    out = ms_deform_attn_core_pytorch(value, shapes, loc, attn)
    # a second, larger case incl. out-of-range locations (zero padding / skip rule)
    torch.manual_seed(4)
    shapes2 = torch.as_tensor([(5, 8), (5, 8)], dtype=torch.long)
    v2 = torch.randn(2, 80, 8, 32)
    l2 = torch.rand(2, 40, 8, 2, 4, 2) * 1.4 - 0.2
    a2 = torch.softmax(torch.randn(2, 40, 8, 8), -1).view(2, 40, 8, 2, 4)
    o2 = ms_deform_attn_core_pytorch(v2, shapes2, l2, a2)
    np.savez_compressed(os.path.join(HERE, "msda_known_answer.npz"),
                        value=value.numpy(), loc=loc.numpy(), attn=attn.numpy(), out=out.numpy(),
                        shapes=shapes.numpy(), value2=v2.numpy(), loc2=l2.numpy(), attn2=a2.numpy(),
                        out2=o2.numpy(), shapes2=shapes2.numpy())


def dump_specs():
    for name in ("unicorn_track_tiny", "unicorn_track_tiny_mask", "unicorn_track_large",
                 "unicorn_track_large_mask", "unicorn_track_large_mot_challenge"):
        m, _ = rb.build_reference_model(name)
        spec = {k: list(v.shape) for k, v in m.named_parameters()}
        with open(os.path.join(HERE, "state_spec_%s.json" % name), "w") as f:
            json.dump(spec, f)


def run_headline():
    """BASELINE.json's headline configuration itself (exp/unicorn_track.py:104: test_size (800, 1280)) through the REAL reference on CPU:
    50 x 80 token grid, pos-embed resized UP (40 -> 50 / 80), the 16000 x 16000 correlation, C = 1536 25 x 40 maps.  ~1.5 min and ~6 GB each."""
    run_sot("unicorn_track_large", 800, 1280, max_full=6144, top_rows=500)
    run_sot("unicorn_track_large_mask", 800, 1280, max_full=6144, top_rows=500)


if __name__ == "__main__":
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "headline":
        run_headline()
        sys.exit(0)
    run_msda_known_answer()
    dump_specs()
    run_sot("unicorn_track_tiny", 320, 320)
    run_sot("unicorn_track_tiny_mask", 320, 320)
    # the headline model and its mask / MOT-challenge (num_classes = 1) variants: the real reference on CPU at 320x320
    run_sot("unicorn_track_large", 320, 320)
    run_sot("unicorn_track_large_mask", 320, 320)
    run_sot("unicorn_track_large_mot_challenge", 320, 320)
    run_sot("unicorn_track_tiny_mask", 320, 512)       # non-square (H / W = 0.625 like 800 x 1280): row / column order of pos-embed, reference points, grids
    run_vos("unicorn_track_tiny_mask", 320, 320)
    run_vos("unicorn_track_large_mask", 320, 320)
    run_headline()
