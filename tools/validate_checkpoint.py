"""One-command validation of a REAL checkpoint on the HIP path (VERDICT r04 "missing" #2).

    python tools/validate_checkpoint.py --ckpt latest_ckpt.pth --exp unicorn_track_large_mask
                                        [--ref /path/to/Unicorn] [--images f0.npy f1.npy ...] [--size 800x1280] [--frames 2]
                                        [--precision f16x2] [--cpu-only] [--json report.json]

TEST / VALIDATION INFRASTRUCTURE, not product code: it imports oracle/ (the CPU restatement of the reference) as the checker.

What it does, in order:
  1. loads the file the way tools/track.py:186-190 does (`torch.load(map_location="cpu")["model"]`) and holds the state dict to the
     learnable-tensor spec of the experiment (unicorn_amd/utils/checkpoint.py:state_spec = the reference's load_ckpt rule,
     unicorn/utils/checkpoint.py:11-33): missing / shape-mismatched / unexpected names are listed, missing or mismatched ones fail;
  2. builds a clip: `--images` (HxWx3 uint8 RGB .npy arrays, letter-boxed like PreprocessorX.process, unicorn_sot.py:111-123) or the
     synthetic clip of oracle/synth.py at --size; frame 0 is the reference frame with a centred init box (or --box x1,y1,x2,y2);
  3. CPU side: the SOT step (unicorn_sot.py:39-55,78-108) on the oracle with the checkpoint's weights; with --ref ALSO on the real
     reference modules (oracle/ref_bootstrap.py, strict load) and the oracle is held to the reference on THIS checkpoint (<= 2e-4);
  4. GPU side (skipped by --cpu-only): the same frames through unicorn_amd (uni_* C-ABI) in saturation check mode
     (`model.check_saturation()`: every f16x2 operand buffer is scanned for the +-65504 bound);
  5. prints the bench-style parity block (box IoU min over the top-500 anchors and over every anchor with a box >= 8 px, SOT decision
     box, embedding cosine min, prior max-abs, CondInst mask IoU for mask models), per-stage rel-L2 and `saturation_stats()`.
Exit status 0 only if the spec check passes, the north_star bar holds (box / mask IoU >= 0.999, embedding cosine within 1e-4), no
operand saturated, and (with --ref) the oracle agrees with the reference.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def parse(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--ckpt", required=True, help="checkpoint file: {'model': state_dict, ...} or a bare state_dict")
    ap.add_argument("--exp", required=True, help="experiment name, e.g. unicorn_track_large_mask")
    ap.add_argument("--ref", default=None, help="root of a MasterBin-IIAU/Unicorn checkout: also run the REAL reference modules on the CPU")
    ap.add_argument("--images", nargs="*", default=None, help="HxWx3 uint8 RGB .npy frames (first = reference frame); default: synthetic clip")
    ap.add_argument("--box", default=None, help="init box x1,y1,x2,y2 in pixels of the first image (default with --images: the centred box [w/4, h/4, 3w/4, 3h/4]; synthetic clip: the box of oracle/synth.py)")
    ap.add_argument("--size", default="800x1280", help="network input HxW")
    ap.add_argument("--frames", type=int, default=2, help="current frames to check (after the reference frame)")
    ap.add_argument("--precision", default="f16x2", choices=["f16x2", "fp32"])
    ap.add_argument("--cpu-only", action="store_true", help="no GPU: spec check + oracle (+ reference with --ref) only")
    ap.add_argument("--threads", type=int, default=min(16, os.cpu_count() or 1))
    ap.add_argument("--json", default=None, help="also write the report to this file")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ helpers
def rel_l2(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def cos_min(a, b):
    a, b = a.detach().double().cpu().flatten(2)[0], b.detach().double().cpu().flatten(2)[0]
    return float(((a * b).sum(0) / (a.norm(dim=0) * b.norm(dim=0)).clamp_min(1e-30)).min())


def box_iou_pairs(a, b):
    """cxcywh rows -> IoU of matching rows"""
    ax1, ay1, ax2, ay2 = a[:, 0] - a[:, 2] / 2, a[:, 1] - a[:, 3] / 2, a[:, 0] + a[:, 2] / 2, a[:, 1] + a[:, 3] / 2
    bx1, by1, bx2, by2 = b[:, 0] - b[:, 2] / 2, b[:, 1] - b[:, 3] / 2, b[:, 0] + b[:, 2] / 2, b[:, 1] + b[:, 3] / 2
    iw = (torch.min(ax2, bx2) - torch.max(ax1, bx1)).clamp(min=0)
    ih = (torch.min(ay2, by2) - torch.max(ay1, by1)).clamp(min=0)
    inter = iw * ih
    return inter / (a[:, 2] * a[:, 3] + b[:, 2] * b[:, 3] - inter)


def _torch_load(path):
    """tensors-only unpickling first (a checkpoint is a user-supplied file: the full unpickler executes arbitrary code); the released
    files hold plain tensors + python scalars, so this succeeds for them; anything else falls back to the reference's own call with a note"""
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:      # noqa: BLE001 -- older files with pickled objects
        print("note: weights_only load of %s failed (%s); falling back to the full unpickler like tools/track.py:186" % (path, type(e).__name__), file=sys.stderr)
        return torch.load(path, map_location="cpu", weights_only=False)


def load_state(path):
    """tools/track.py:186-188"""
    ckpt = _torch_load(path)
    sd = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt and isinstance(ckpt["model"], dict) else ckpt
    return {k: (v.float() if torch.is_floating_point(v) else v) for k, v in sd.items() if torch.is_tensor(v)}


def spec_report(exp, sd):
    from unicorn_amd.utils.checkpoint import filter_ckpt, state_spec
    spec = state_spec(exp)
    _, missing, mismatched = filter_ckpt(spec, sd)
    unexpected = [k for k in sd if k not in spec]
    nonfinite = [k for k in spec if k in sd and not bool(torch.isfinite(sd[k]).all())]
    return {"tensors_expected": len(spec), "missing": missing, "shape_mismatch": mismatched, "unexpected": unexpected[:20],
            "unexpected_count": len(unexpected), "non_finite": nonfinite,
            "max_abs_weight": max((float(sd[k].abs().max()) for k in spec if k in sd and sd[k].numel()), default=0.0),
            "ok": not missing and not mismatched and not nonfinite}


def make_clip(args, H, W):
    """-> frames [(1,3,H,W) float BGR 0-255], init box xyxy in network pixels"""
    if args.images:
        import letterbox_oracle as lo
        frames, r = [], 1.0
        for f in args.images[:1 + args.frames]:
            img = np.load(f)
            assert img.ndim == 3 and img.shape[2] == 3 and img.dtype == np.uint8, "%s: HxWx3 uint8 expected" % f
            out, r = lo.letterbox(img, (H, W), swap_rb=True)          # PreprocessorX.process
            frames.append(torch.from_numpy(np.ascontiguousarray(out)).float().view(1, 3, H, W))
        h0, w0 = np.load(args.images[0]).shape[:2]
        box = torch.tensor([float(v) for v in args.box.split(",")]) if args.box else torch.tensor([w0 / 4, h0 / 4, 3 * w0 / 4, 3 * h0 / 4])      # centred, half the image per side
        return frames, box * r
    import synth
    frames, box = synth.synth_clip(H, W, 1 + args.frames, seed=1)
    if args.box:
        box = torch.tensor([float(v) for v in args.box.split(",")])
    return frames, box


# ------------------------------------------------------------------------------------------------ the three implementations
def run_oracle(P, cfg, frames, box):
    import unicorn_oracle as uo
    with torch.no_grad():
        st = uo.sot_init(P, cfg, frames[0], box)
        return st, [uo.sot_step(P, cfg, st, f) for f in frames[1:]]


def run_reference(ref_root, exp, sd, cfg, frames, box):
    """the driver lines unicorn_sot.py:39-55,78-108 on the REAL reference nn.Module (CPU; `.cuda()` / fp16 casts dropped)"""
    os.environ["UNICORN_REFERENCE"] = ref_root
    import ref_bootstrap as rb
    rb.REF_ROOT = ref_root
    import torch.nn.functional as F
    model, _ = rb.build_reference_model(exp)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert all("mask_head" in m or m.endswith("_iter") for m in missing), ("reference model misses", missing[:8])
    H, W = frames[0].shape[-2:]
    outs = []
    with torch.no_grad():
        _, d_pre = model(imgs=frames[0], mode="backbone")
        lab = torch.zeros((1, 1, H, W))
        x1, y1, x2, y2 = torch.round(box).int().tolist()
        lab[0, 0, max(0, min(y1, H)):max(0, min(y2, H)), max(0, min(x1, W)):max(0, min(x2, W))] = 1.0
        lbs = F.interpolate(lab, scale_factor=1 / 8, mode="bilinear", align_corners=False)[0].flatten(-2)
        for img in frames[1:]:
            fpn, d_cur = model(imgs=img, mode="backbone")
            f_pre, f_cur = model(seq_dict0=d_pre, seq_dict1=d_cur, mode="interaction")
            e_pre, e_cur = model(feat=f_pre, mode="upsample"), model(feat=f_cur, mode="upsample")
            trans = torch.softmax(torch.mm(e_pre.flatten(-2).squeeze().transpose(1, 0), e_cur.flatten(-2).squeeze()), dim=0)
            coarse = (lbs @ trans).view(1, -1, d_cur["h"] * 2, d_cur["w"] * 2).float()
            pri = (coarse, F.interpolate(coarse, scale_factor=1 / 2, mode="bilinear", align_corners=False),
                   F.interpolate(coarse, scale_factor=1 / 4, mode="bilinear", align_corners=False))
            head = model.head(fpn, pri, mode="sot")
            outs.append({"fpn": fpn, "seq": d_cur, "feat_pre": f_pre, "feat_cur": f_cur, "embed_pre": e_pre, "embed_cur": e_cur,
                         "coarse": coarse, "head": head})
    return outs


def run_hip(exp, sd, cfg, frames, box, precision):
    from unicorn_amd.models import Unicorn
    from unicorn_amd.ops import condinst_masks, corr_softmax_pv, label_map_s8, prior_pyramid
    from unicorn_amd.utils.checkpoint import load_ckpt
    H, W = frames[0].shape[-2:]
    m = Unicorn(exp, precision=precision).cuda()
    load_ckpt(m, sd)                                                   # the reference's loader rule (checkpoint.py:11-33)
    m.eval()
    m.check_saturation(True)
    outs = []
    with torch.no_grad():
        _, d_pre = m(imgs=frames[0].cuda(), mode="backbone")
        lbs = label_map_s8(box, H, W, "cuda")
        for img in frames[1:]:
            fpn, d_cur = m(imgs=img.cuda(), mode="backbone")
            f_pre, f_cur = m(seq_dict0=d_pre, seq_dict1=d_cur, mode="interaction")
            e_pre, e_cur = m(feat=f_pre, mode="upsample"), m(feat=f_cur, mode="upsample")
            pred = corr_softmax_pv(e_pre.flatten(-2).squeeze(0), e_cur.flatten(-2).squeeze(0), lbs, precision=0 if precision == "fp32" else 2)
            coarse = pred.view(1, -1, d_cur["h"] * 2, d_cur["w"] * 2)
            head = m.head(fpn, prior_pyramid(coarse), mode="sot")
            outs.append({"fpn": fpn, "seq": d_cur, "feat_pre": f_pre, "feat_cur": f_cur, "embed_pre": e_pre, "embed_cur": e_cur,
                         "coarse": coarse, "head": head})
    torch.cuda.synchronize()
    sat = m.saturation_stats()
    return outs, sat, (m, condinst_masks)


# ------------------------------------------------------------------------------------------------ comparison
def compare(got, want, cfg, masks=None):
    """stage metrics of one frame: `got` against `want` (the CPU fp32 side)"""
    import unicorn_oracle as uo
    met = {}
    for i in range(3):
        met["fpn%d" % i] = rel_l2(got["fpn"][i], want["fpn"][i])
    met["seq_feat"] = rel_l2(got["seq"]["feat"], want["seq"]["feat"])
    met["feat_cur"] = rel_l2(got["feat_cur"], want["feat_cur"])
    met["embed_cur"] = rel_l2(got["embed_cur"], want["embed_cur"])
    met["embed_cos_min"] = min(cos_min(got["embed_cur"], want["embed_cur"]), cos_min(got["embed_pre"], want["embed_pre"]))
    met["prior_max_abs"] = float((got["coarse"].cpu() - want["coarse"]).abs().max())
    ho = want["head"][0] if cfg.mask else want["head"]
    hh = (got["head"][0] if cfg.mask else got["head"]).cpu()
    score = ho[0, :, 4] * ho[0, :, 5]
    top = torch.argsort(score, descending=True)[:500]
    met["score_max"] = float(score.max())
    met["box_iou_min_top500"] = float(box_iou_pairs(hh[0, top, :4], ho[0, top, :4]).min())
    iou_all = box_iou_pairs(hh[0, :, :4], ho[0, :, :4])
    big = (ho[0, :, 2] >= 8.0) & (ho[0, :, 3] >= 8.0)
    met["box_iou_min_all_8px"] = float(iou_all[big].min()) if bool(big.any()) else 1.0
    met["score_max_abs_err"] = float((hh[0, :, 4] * hh[0, :, 5] - score).abs().max())
    det_o = uo.postprocess(ho.clone(), 1, 0.001, 0.65)[0]
    det_h = uo.postprocess(hh.clone(), 1, 0.001, 0.65)[0]
    met["sot_detections"] = [0 if det_o is None else int(det_o.shape[0]), 0 if det_h is None else int(det_h.shape[0])]
    if det_o is not None and det_h is not None:
        a, b = det_h[:1, :4], det_o[:1, :4]
        cx = lambda t: torch.stack([(t[:, 0] + t[:, 2]) / 2, (t[:, 1] + t[:, 3]) / 2, t[:, 2] - t[:, 0], t[:, 3] - t[:, 1]], 1)  # noqa: E731
        met["sot_box_iou"] = float(box_iou_pairs(cx(a), cx(b))[0])
    if cfg.mask:
        for n, i in (("dyn_params", 2), ("mask_feats", 4), ("up_masks", 5)):
            met[n] = rel_l2(got["head"][i], want["head"][i])
        if masks is not None:
            met.update(masks(got, want))
    return met


def hip_masks(cfg, condinst_masks):
    import unicorn_oracle as uo

    def f(got, want):
        ohead = tuple(t.clone() for t in want["head"])
        det, idx = uo.postprocess(ohead[0], 1, 0.001, 0.65, return_index=True)[0]
        if det is None:
            return {}
        idx = idx[:16]
        mo = uo.aligned_bilinear(uo.dynamic_mask_head(cfg, want["head"][4], want["head"][2][0][idx], want["head"][1][idx],
                                                      want["head"][3][0][idx], want["head"][5]), cfg.d_rate)
        mh = condinst_masks(got["head"][4], got["head"][5], got["head"][2][0][idx.cuda()], got["head"][1][idx.cuda()],
                            got["head"][3][0][idx], cfg.up_rate, cfg.d_rate).cpu()
        a, b = mh > 0.5, mo > 0.5
        inter = (a & b).flatten(1).sum(1).float()
        union = (a | b).flatten(1).sum(1).float().clamp_min(1)
        return {"mask_iou_min": float((inter / union).min()), "masks_checked": int(idx.numel())}
    return f


def bar_ok(mets):
    ok = True
    for m in mets:
        ok &= m["box_iou_min_top500"] >= 0.999 and m["box_iou_min_all_8px"] >= 0.999 and m["embed_cos_min"] >= 1 - 1e-4
        ok &= m.get("sot_box_iou", 1.0) >= 0.999 and m.get("mask_iou_min", 1.0) >= 0.999
    return bool(ok)


def main(argv=None):
    args = parse(argv)
    import unicorn_oracle as uo
    H, W = (int(v) for v in args.size.lower().split("x"))
    assert H % 32 == 0 and W % 32 == 0, "network input must be a multiple of 32"
    cfg = uo.CONFIGS[args.exp]
    torch.set_num_threads(args.threads)
    sd = load_state(args.ckpt)
    report = {"ckpt": args.ckpt, "exp": args.exp, "size": [H, W], "frames": args.frames, "precision": args.precision,
              "spec": spec_report(args.exp, sd)}
    ok = report["spec"]["ok"]
    if ok:
        frames, box = make_clip(args, H, W)
        report["clip"] = "images" if args.images else "synthetic (oracle/synth.py, seed 1)"
        st, orc = run_oracle(sd, cfg, frames, box)
        report["oracle_finite"] = all(bool(torch.isfinite(o["head"][0] if cfg.mask else o["head"]).all()) for o in orc)
        ok &= report["oracle_finite"]
        if args.ref:
            ref = run_reference(args.ref, args.exp, sd, cfg, frames, box)
            mets = [compare(o, r, cfg) for o, r in zip(orc, ref)]
            report["oracle_vs_reference"] = {"per_frame": mets, "tolerance": "rel-L2 <= 2e-4 per stage, north_star bar on boxes / cosine"}
            good = bar_ok(mets) and all(max(m[k] for k in ("fpn0", "fpn1", "fpn2", "seq_feat", "feat_cur", "embed_cur")) <= 2e-4 for m in mets)
            report["oracle_vs_reference"]["pass"] = bool(good)
            ok &= good
        if not args.cpu_only:
            if not torch.cuda.is_available():
                raise SystemExit("validate_checkpoint: no GPU (use --cpu-only for the spec / oracle / reference checks)")
            hip, sat, (m, cm) = run_hip(args.exp, sd, cfg, frames, box, args.precision)
            mets = [compare(h, o, cfg, hip_masks(cfg, cm) if cfg.mask else None) for h, o in zip(hip, orc)]
            parity = {"vs": "CPU oracle (fp32) with the checkpoint's weights, %d frame(s)" % len(mets),
                      "box_iou_min": min(x["box_iou_min_top500"] for x in mets), "box_iou_min_all_8px": min(x["box_iou_min_all_8px"] for x in mets),
                      "embed_cos_min": min(x["embed_cos_min"] for x in mets), "prior_max_abs": max(x["prior_max_abs"] for x in mets),
                      "mask_iou_min": min((x["mask_iou_min"] for x in mets if "mask_iou_min" in x), default=None),
                      "sot_box_iou_min": min((x["sot_box_iou"] for x in mets if "sot_box_iou" in x), default=None),
                      "bar": "box/mask IoU >= 0.999, embedding cosine within 1e-4 (BASELINE.json north_star)", "pass": bar_ok(mets)}
            report["hip_vs_oracle"] = {"per_frame": mets}
            report["parity"] = parity
            report["saturation"] = dict(sat, bound=65504.0, ok=sat["saturated"] == 0)
            ok &= parity["pass"] and sat["saturated"] == 0
    report["pass"] = bool(ok)
    txt = json.dumps(report, indent=1)
    print(txt)
    if args.json:
        with open(args.json, "w") as f:
            f.write(txt)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
