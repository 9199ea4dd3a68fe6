"""Per-stage precision budget of the hot path (TEST INFRASTRUCTURE / analysis, CPU only).

Question: which layers move box IoU when GEMM operands are narrower than fp32, and
which operand format is the cheapest that keeps `box_iou_min_top500 >= 0.999` against the fp32 oracle?

Method: the oracle's F.conv2d / F.linear are wrapped so that, per stage, BOTH operands are rounded to a
chosen format before the (fp32-accumulated) contraction - exactly what an MFMA on split operands computes:
  bf16     8-bit mantissa                     1 MFMA per product
  f16      11-bit mantissa                    1 MFMA
  bf16x2   hi + lo bf16 pieces (16 bits)      3 MFMAs (hi*hi, hi*lo, lo*hi)
  f16x2    hi + lo f16 pieces (22 bits)       3 MFMAs
  bf16x3   3 bf16 pieces (24 bits)            6 MFMAs
Depthwise convs (groups > 1) are VALU fp32 in the HIP path and stay exact here.

    python oracle/error_budget.py [--model unicorn_track_tiny] [--size 320 320]
"""
import argparse
import json
import os
import sys
import types

import torch
import torch.nn.functional as RF

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth  # noqa: E402
import unicorn_oracle as uo  # noqa: E402


def q_bf16(x):
    return x.bfloat16().float()


def q_f16(x):
    return x.half().float()


def q_bf16x2(x):
    hi = x.bfloat16().float()
    return hi + (x - hi).bfloat16().float()


def q_bf16x3(x):
    hi = x.bfloat16().float()
    r = x - hi
    mid = r.bfloat16().float()
    return hi + mid + (r - mid).bfloat16().float()


def q_f16x2(x):
    hi = x.half().float()
    return hi + (x - hi).half().float()


QUANT = {"fp32": lambda x: x, "bf16": q_bf16, "f16": q_f16, "bf16x2": q_bf16x2, "f16x2": q_f16x2, "bf16x3": q_bf16x3}
STAGES = ("backbone", "fpn", "interaction", "upsample", "head")


class Policy:
    def __init__(self, default="fp32", **per_stage):
        self.fmt = {s: per_stage.get(s, default) for s in STAGES}
        self.stage = "backbone"

    def q(self, x):
        return QUANT[self.fmt[self.stage]](x)


POLICY = Policy()


def _conv2d(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    if groups == 1:
        x, w = POLICY.q(x), POLICY.q(w)
    return RF.conv2d(x, w, b, stride, padding, dilation, groups)


def _linear(x, w, b=None):
    return RF.linear(POLICY.q(x), POLICY.q(w), b)


class _FProxy(types.ModuleType):
    def __getattr__(self, k):
        return getattr(RF, k)


def install():
    fp = _FProxy("F_quant")
    fp.conv2d = _conv2d
    fp.linear = _linear
    uo.F = fp
    # stage tags
    for name, stage in (("convnext_features", "backbone"), ("pafpn", "fpn"), ("forward_interaction", "interaction"),
                        ("forward_upsample", "upsample"), ("_head_trunk", "head"), ("mask_branch", "head")):
        orig = getattr(uo, name)

        def wrap(*a, _o=orig, _s=stage, **k):
            prev, POLICY.stage = POLICY.stage, _s
            try:
                return _o(*a, **k)
            finally:
                POLICY.stage = prev
        setattr(uo, name, wrap)


def box_iou_pairs(a, b):
    ax1, ay1, ax2, ay2 = a[:, 0] - a[:, 2] / 2, a[:, 1] - a[:, 3] / 2, a[:, 0] + a[:, 2] / 2, a[:, 1] + a[:, 3] / 2
    bx1, by1, bx2, by2 = b[:, 0] - b[:, 2] / 2, b[:, 1] - b[:, 3] / 2, b[:, 0] + b[:, 2] / 2, b[:, 1] + b[:, 3] / 2
    iw = (torch.min(ax2, bx2) - torch.max(ax1, bx1)).clamp(min=0)
    ih = (torch.min(ay2, by2) - torch.max(ay1, by1)).clamp(min=0)
    inter = iw * ih
    return inter / (a[:, 2] * a[:, 3] + b[:, 2] * b[:, 3] - inter)


def run(P, cfg, frames, box, policy):
    global POLICY
    POLICY = policy
    with torch.no_grad():
        st = uo.sot_init(P, cfg, frames[0], box)
        return uo.sot_step(P, cfg, st, frames[1])


def metrics(o, ref, cfg):
    rel = lambda a, b: float((a - b).norm() / b.norm())
    ho = ref["head"][0] if cfg.mask else ref["head"]
    hh = o["head"][0] if cfg.mask else o["head"]
    score = ho[0, :, 4] * ho[0, :, 5]
    top = torch.argsort(score, descending=True)[:500]
    iou = box_iou_pairs(hh[0, top, :4], ho[0, top, :4])
    ea, eb = o["embed_cur"].flatten(2)[0].double(), ref["embed_cur"].flatten(2)[0].double()
    cos = (ea * eb).sum(0) / (ea.norm(dim=0) * eb.norm(dim=0))
    m = {"seq_feat": rel(o["seq"]["feat"], ref["seq"]["feat"]), "fpn0": rel(o["fpn"][0], ref["fpn"][0]),
         "fpn2": rel(o["fpn"][2], ref["fpn"][2]), "embed": rel(o["embed_cur"], ref["embed_cur"]),
         "embed_cos_min": float(cos.min()), "prior_maxabs": float((o["coarse"] - ref["coarse"]).abs().max()),
         "iou_min": float(iou.min()), "iou_mean": float(iou.mean()),
         "score_rel": float(((hh[0, top, 4] * hh[0, top, 5] - score[top]).abs() / score[top]).max())}
    if cfg.mask:
        m["dyn"] = rel(o["head"][2], ref["head"][2])
        m["mask_feats"] = rel(o["head"][4], ref["head"][4])
    return m


def trained_like(cfg, seed):
    """Synthetic weights shaped like a TRAINED detector's (no checkpoint exists offline): the synthetic draw of oracle/synth.py with
    (a) obj / cls prediction biases planted at logit(0.5) + N(0, 1) per level instead of -4.5 (scores spread over 0.05 - 0.95, so the
    top-500 anchors are confident detections, not the tail of a 1e-4 distribution), (b) the last prediction layers (reg / obj / cls
    preds) damped x 0.25 (a trained regressor's output varies smoothly with its input; an undamped random one amplifies feature noise),
    (c) a different seed per draw."""
    P = synth.synth_state_dict(cfg, seed=100 + seed)
    g = torch.Generator().manual_seed(seed)
    for k in list(P):
        if k.startswith("head.") and any(t in k for t in ("cls_preds", "obj_preds", "reg_preds")):
            if k.endswith("weight"):
                P[k] = P[k] * 0.25
            elif "reg_preds" not in k:
                P[k] = torch.randn(P[k].shape, generator=g)
    return P


class AsymPolicy:
    """per stage a PAIR (activation format, weight format): the asymmetric two-MFMA candidates of round 6 -- "a22w11" = f16x2 activations x f16
    weights (a_hi w_hi + a_lo w_hi), "a11w22" = f16 activations x f16x2 weights (a_hi w_hi + a_hi w_lo) -- against the symmetric f16x2 (3 MFMAs)"""

    def __init__(self, default=("f16x2", "f16x2"), **per_stage):
        self.fmt = {s: per_stage.get(s, default) for s in STAGES}
        self.stage = "backbone"

    def qa(self, x):
        return QUANT[self.fmt[self.stage][0]](x)

    def qw(self, w):
        return QUANT[self.fmt[self.stage][1]](w)

    q = qa


def run_asymmetric(cfg, out_path, seeds=2, H=320, W=320):
    """profiles/r06_precision_budget_asymmetric_*.json: uniform and single-stage a22w11 / a11w22 policies on the synthetic draw and on the
    trained-like ensemble.  Result (tiny, 320 x 320): every two-MFMA policy on backbone, FPN or head breaks the box-IoU bar (0.925-0.998); on
    interaction / upsample alone it passes with trained-like heads (0.9990-0.9998) and FAILS with the synthetic draw (0.9954-0.9982) -- the parity
    tests run on the synthetic draw, and those two stages are ~5 % of the frame: three MFMAs per product stay."""
    def conv2d(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        if groups == 1:
            x, w = POLICY.qa(x), POLICY.qw(w)
        return RF.conv2d(x, w, b, stride, padding, dilation, groups)
    uo.F.conv2d = conv2d
    uo.F.linear = lambda x, w, b=None: RF.linear(POLICY.qa(x), POLICY.qw(w), b)
    fp = AsymPolicy(("fp32", "fp32"))
    pols = {"a22w22": AsymPolicy(), "a22w11": AsymPolicy(("f16x2", "f16")), "a11w22": AsymPolicy(("f16", "f16x2")), "a11w11": AsymPolicy(("f16", "f16"))}
    for st in STAGES:
        pols["a22w11@" + st] = AsymPolicy(**{st: ("f16x2", "f16")})
        pols["a11w22@" + st] = AsymPolicy(**{st: ("f16", "f16x2")})
    out = {}
    draws = [("synthetic", synth.synth_state_dict(cfg), 1)] + [("trained_like_%d" % s_, trained_like(cfg, s_), 1 + s_) for s_ in range(seeds)]
    for tag, P, cs in draws:
        frames, box = synth.synth_clip(H, W, 2, seed=cs)
        ref = run(P, cfg, frames, box, fp)
        out[tag] = {}
        for name, pol in pols.items():
            m = metrics(run(P, cfg, frames, box, pol), ref, cfg)
            out[tag][name] = m
            print("%-16s %-22s iou_min %.6f iou_mean %.6f embed_cos_min %.7f" % (tag, name, m["iou_min"], m["iou_mean"], m["embed_cos_min"]), flush=True)
    if out_path:
        json.dump(out, open(out_path, "w"), indent=1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="unicorn_track_tiny")
    ap.add_argument("--size", type=int, nargs=2, default=[320, 320])
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--out", default=None)
    ap.add_argument("--quick", action="store_true", help="uniform policies only")
    ap.add_argument("--ensemble", type=int, default=0, metavar="N",
                    help="N weight draws of a TRAINED-LIKE ensemble instead of the single synthetic draw (VERDICT r03 weak #3): head biases "
                         "planted so that obj * cls scores are O(0.1 - 0.9) instead of ~1e-4, regression / prediction weights damped so the "
                         "top anchors carry boxes of the init-box scale, different seeds; uniform policies only")
    ap.add_argument("--asymmetric", action="store_true", help="two-MFMA candidates: f16x2 activations x f16 weights and the reverse, uniform and per stage")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    install()
    cfg = uo.CONFIGS[args.model]
    if args.asymmetric:
        run_asymmetric(cfg, args.out, H=args.size[0], W=args.size[1])
        return
    P = synth.synth_state_dict(cfg)
    H, W = args.size
    frames, box = synth.synth_clip(H, W, 2, seed=1)
    if args.ensemble:
        rows = {}
        for seed in range(args.ensemble):
            P = trained_like(cfg, seed)
            frames, box = synth.synth_clip(H, W, 2, seed=1 + seed)
            ref = run(P, cfg, frames, box, Policy("fp32"))
            ho = ref["head"][0] if cfg.mask else ref["head"]
            sc = (ho[0, :, 4] * ho[0, :, 5]).sort(descending=True)[0]
            print("seed %d: top score %.3f, 500th %.3f" % (seed, float(sc[0]), float(sc[499])), flush=True)
            for f in ("bf16", "f16", "bf16x2", "f16x2"):
                m = metrics(run(P, cfg, frames, box, Policy(f)), ref, cfg)
                rows["seed%d all=%s" % (seed, f)] = m
                print("seed", seed, f, json.dumps({k: round(v, 6) for k, v in m.items() if k in ("iou_min", "iou_mean", "embed_cos_min", "fpn0", "score_rel")}), flush=True)
        summ = {f: {"iou_min": min(rows["seed%d all=%s" % (s_, f)]["iou_min"] for s_ in range(args.ensemble)),
                    "iou_mean_min": min(rows["seed%d all=%s" % (s_, f)]["iou_mean"] for s_ in range(args.ensemble))}
                for f in ("bf16", "f16", "bf16x2", "f16x2")}
        print("summary", json.dumps(summ))
        if args.out:
            json.dump({"model": args.model, "size": [H, W], "ensemble": args.ensemble, "what": trained_like.__doc__, "summary": summ, "rows": rows},
                      open(args.out, "w"), indent=1)
        return
    ref = run(P, cfg, frames, box, Policy("fp32"))
    rows = {}

    def go(tag, pol):
        rows[tag] = metrics(run(P, cfg, frames, box, pol), ref, cfg)
        print(tag, json.dumps(rows[tag]), flush=True)

    for f in ("bf16", "f16", "bf16x2", "f16x2", "bf16x3"):
        go("all=" + f, Policy(f))
    if not args.quick:
        for s in STAGES:                                      # one stage narrow, rest exact: who moves IoU?
            go("only_%s=bf16" % s, Policy("fp32", **{s: "bf16"}))
        for s in STAGES:
            go("only_%s=f16" % s, Policy("fp32", **{s: "f16"}))
        go("bf16_backbone+bf16x2_rest", Policy("bf16x2", backbone="bf16"))
        go("bf16x2_all_but_head_f16x2", Policy("bf16x2", head="f16x2"))
    if args.out:
        json.dump({"model": args.model, "size": [H, W], "rows": rows}, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
