"""End-to-end parity of the HIP path (through unicorn_amd's reference-shaped Python API -> C-ABI) against
(a) golden vectors produced by the REAL reference (tiny, 320x320) and (b) the CPU oracle at 800x1280.

Three precision modes of the same kernels are tested (DESIGN.md "precision"):
  precision="f16x2" fp32-equivalent split-f16 MFMA operands (the bench headline) and
  precision="fp32"  exact-fp32 MFMA everywhere: both must meet the north_star bar
        box IoU >= 0.999, mask IoU >= 0.999, embedding cosine within 1e-4 (min over pixels >= 1 - 1e-4),
        feature maps rel-L2 <= 1e-4, propagated prior max-abs <= 1e-4
  precision="bf16"  RETIRED secondary mode (bf16 MFMA operands; cannot meet the bar with any weights): one
        320x320 golden case keeps its kernels exercised (feature maps rel-L2 <= 5e-2, prior max-abs <= 2e-2, finite boxes).
The measured values are written to gpurun_out/parity_metrics.json."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import synth  # noqa: E402
import unicorn_oracle as uo  # noqa: E402

MAX_FULL = 1 << 15
METRICS = {}
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


_SESSION_START = __import__("time").time()


def _dump():
    """merge into gpurun_out/parity_metrics.json: sub-processes of the suite (tools/validate_checkpoint.py, bench.py) and a second pytest process write
    the same file; only a file older than this session is replaced"""
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", "parity_metrics.json")
    old = {}
    try:
        if os.path.getmtime(path) >= _SESSION_START - 1.0:
            old = json.load(open(path))
    except (OSError, ValueError):
        old = {}
    old.update(METRICS)
    with open(path, "w") as f:
        json.dump(old, f, indent=1)


def sample(t):
    a = t.detach().float().cpu().contiguous().numpy().reshape(-1)
    if a.size <= MAX_FULL:
        return a
    return a[np.linspace(0, a.size - 1, MAX_FULL).astype(np.int64)]


def rel_l2(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def cos_per_pixel(a, b):
    """a,b (1,C,H,W) -> per-pixel cosine"""
    a, b = a.float().cpu().flatten(2)[0].double(), b.float().cpu().flatten(2)[0].double()
    return ((a * b).sum(0) / (a.norm(dim=0) * b.norm(dim=0)).clamp_min(1e-30)).numpy()


def box_iou_pairs(a, b):
    """cxcywh rows -> IoU of matching rows"""
    ax1, ay1, ax2, ay2 = a[:, 0] - a[:, 2] / 2, a[:, 1] - a[:, 3] / 2, a[:, 0] + a[:, 2] / 2, a[:, 1] + a[:, 3] / 2
    bx1, by1, bx2, by2 = b[:, 0] - b[:, 2] / 2, b[:, 1] - b[:, 3] / 2, b[:, 0] + b[:, 2] / 2, b[:, 1] + b[:, 3] / 2
    iw = (torch.min(ax2, bx2) - torch.max(ax1, bx1)).clamp(min=0)
    ih = (torch.min(ay2, by2) - torch.max(ay1, by1)).clamp(min=0)
    inter = iw * ih
    return inter / (a[:, 2] * a[:, 3] + b[:, 2] * b[:, 3] - inter)


def build(name, precision="bf16"):
    from unicorn_amd.models import Unicorn
    cfg = uo.CONFIGS[name]
    P = synth.synth_state_dict(cfg)
    m = Unicorn(name, precision=precision).cuda()
    missing, unexpected = m.load_state_dict(P, strict=False)
    assert not missing, missing[:5]
    return m.eval(), cfg, P


def hip_sot_step(m, cfg, frames, box):
    """the driver logic of external/lib/test/tracker/unicorn_sot.py:39-55,78-108 on the drop-in API"""
    from unicorn_amd.ops import corr_softmax_pv, label_map_s8, prior_pyramid
    H, W = frames[0].shape[-2:]
    with torch.no_grad():
        _, d_pre = m(imgs=frames[0].cuda(), mode="backbone")
        lbs = label_map_s8(box, H, W, "cuda")
        fpn, d_cur = m(imgs=frames[1].cuda(), mode="backbone")
        f_pre, f_cur = m(seq_dict0=d_pre, seq_dict1=d_cur, mode="interaction")
        e_pre, e_cur = m(feat=f_pre, mode="upsample"), m(feat=f_cur, mode="upsample")
        pred = corr_softmax_pv(e_pre.flatten(-2).squeeze(0), e_cur.flatten(-2).squeeze(0), lbs)
        coarse = pred.view(1, -1, d_cur["h"] * 2, d_cur["w"] * 2)
        pri = prior_pyramid(coarse)
        head = m.head(fpn, pri, mode="sot")
    torch.cuda.synchronize()
    return dict(lbs=lbs, fpn=fpn, seq=d_cur, feat_pre=f_pre, feat_cur=f_cur, embed_pre=e_pre, embed_cur=e_cur, coarse=coarse,
                pri=pri, head=head)


EXACT = ("fp32", "f16x2")      # precision modes that must meet the north_star bar


# bf16 is a RETIRED secondary mode (it cannot meet the north_star bar with any weights): ONE golden case keeps its
# kernels building and finite; the 800x1280 / ragged / batched bf16 runs of rounds 1-4 were dropped from the suite in round 5
GOLDEN_CASES = [(e, p, 320, 320) for e in ("unicorn_track_tiny", "unicorn_track_tiny_mask") for p in ("fp32", "f16x2")] + \
    [("unicorn_track_tiny", "bf16", 320, 320)] + \
    [(e, "f16x2", 320, 320) for e in ("unicorn_track_large", "unicorn_track_large_mask", "unicorn_track_large_mot_challenge")] + \
    [("unicorn_track_tiny_mask", "f16x2", 320, 512)]      # non-square golden of the real reference (the 800 x 1280 aspect)


@pytest.mark.parametrize("exp,precision,H,W", GOLDEN_CASES)
def test_tiny_320_vs_reference_golden(exp, precision, H, W, golden_dir):
    """BASELINE.json configs[0] shapes; expected values come from the real reference (tests/golden/make_golden.py): the tiny
    models in all three precisions, the headline `unicorn_track_large` family in the headline precision, and one non-square map."""
    g = np.load(os.path.join(golden_dir, "%s_%dx%d.npz" % (exp, H, W)))
    m, cfg, P = build(exp, precision)
    frames, box = synth.synth_clip(H, W, 2, seed=1)
    r = hip_sot_step(m, cfg, frames, box)
    met = {}
    assert np.array_equal(sample(r["lbs"]), g["lbs_pre"])
    for i in range(3):
        met["fpn%d" % i] = rel_l2(sample(r["fpn"][i]), g["fpn%d" % i])
    met["seq_feat"] = rel_l2(sample(r["seq"]["feat"]), g["seq_feat"])
    met["seq_pos"] = rel_l2(sample(r["seq"]["pos"]), g["seq_pos"])
    met["feat_pre"] = rel_l2(sample(r["feat_pre"]), g["feat_pre"])
    met["feat_cur"] = rel_l2(sample(r["feat_cur"]), g["feat_cur"])
    met["embed_pre"] = rel_l2(sample(r["embed_pre"]), g["embed_pre"])
    met["embed_cur"] = rel_l2(sample(r["embed_cur"]), g["embed_cur"])
    met["coarse_maxabs"] = float(np.abs(sample(r["coarse"]) - g["coarse"]).max())
    head = r["head"][0] if cfg.mask else r["head"]
    hg = torch.from_numpy(g["head_out"]).reshape(tuple(g["head_out__shape"]))[0]
    hh = head[0].cpu()
    score = hg[:, 4] * hg[:, 5]
    top = torch.argsort(score, descending=True)[:200]
    iou = box_iou_pairs(hh[top, :4], hg[top, :4])
    met["box_iou_min_top200"] = float(iou.min())
    met["box_iou_mean_top200"] = float(iou.mean())
    met["score_relerr_top200"] = float(((hh[top, 4] * hh[top, 5] - score[top]).abs() / score[top]).max())
    if cfg.mask:
        for n, t in zip(["dyn_params", "mask_feats", "up_masks"], [r["head"][2], r["head"][4], r["head"][5]]):
            met[n] = rel_l2(sample(t), g[n])
        assert np.allclose(sample(r["head"][1]), g["locations"])
        assert np.array_equal(sample(r["head"][3]), g["fpn_levels"])
    METRICS["golden_%s_%s%s" % (exp, precision, "" if (H, W) == (320, 320) else "_%dx%d" % (H, W))] = met
    _dump()
    assert met["seq_pos"] < 1e-5
    feat_tol, prior_tol = (1e-4, 1e-4) if precision in EXACT else (5e-2, 2e-2)
    for k in ("fpn0", "fpn1", "fpn2", "seq_feat", "feat_pre", "feat_cur", "embed_pre", "embed_cur"):
        assert met[k] < feat_tol, (k, met[k])
    assert met["coarse_maxabs"] < prior_tol
    if precision in EXACT:
        assert met["box_iou_min_top200"] > 0.999, met
        assert met["score_relerr_top200"] < 1e-3
        if cfg.mask:
            assert max(met["dyn_params"], met["mask_feats"], met["up_masks"]) < 1e-4, met
    else:
        assert met["box_iou_mean_top200"] > 0.75, met


@pytest.mark.parametrize("exp", ["unicorn_track_large", "unicorn_track_large_mask"])
def test_headline_800x1280_vs_reference_golden(exp, golden_dir):
    """The HEADLINE configuration itself (exp/unicorn_track.py:104: test_size (800, 1280), `unicorn_track_large` and its mask variant)
    against vectors of the REAL reference run on the CPU at that size (tests/golden/make_golden.py:run_headline): the 50 x 80 token grid,
    the pos-embed resized UP (40 -> 50 / 80), the 16000 x 16000 correlation and the C = 1536 25 x 40 maps are compared with the reference,
    not only with the oracle.  Feature maps on the stored strided sample (`__max_full` per tensor), boxes on the 500 best raw head rows
    (SOT head and mode="whole"), dynamic parameters of the 64 best, CondInst mask bits of the reference's first detections."""
    H, W = 800, 1280
    g = np.load(os.path.join(golden_dir, "%s_%dx%d.npz" % (exp, H, W)))
    mf = int(g["__max_full"][0])

    def smp(t):
        a = t.detach().float().cpu().contiguous().numpy().reshape(-1)
        return a if a.size <= mf else a[np.linspace(0, a.size - 1, mf).astype(np.int64)]

    m, cfg, P = build(exp, "f16x2")
    frames, box = synth.synth_clip(H, W, 2, seed=1)
    r = hip_sot_step(m, cfg, frames, box)
    met = {}
    assert np.array_equal(smp(r["lbs"]), g["lbs_pre"])
    for k, t in (("fpn0", r["fpn"][0]), ("fpn1", r["fpn"][1]), ("fpn2", r["fpn"][2]), ("seq_feat", r["seq"]["feat"]), ("seq_pos", r["seq"]["pos"]),
                 ("feat_pre", r["feat_pre"]), ("feat_cur", r["feat_cur"]), ("embed_pre", r["embed_pre"]), ("embed_cur", r["embed_cur"])):
        assert tuple(g[k + "__shape"]) == tuple(t.shape), k
        met[k] = rel_l2(smp(t), g[k])
    met["coarse_maxabs"] = float(np.abs(smp(r["coarse"]) - g["coarse"]).max())
    met["prior16_maxabs"] = float(np.abs(smp(r["pri"][1]) - g["prior16"]).max())
    met["prior32_maxabs"] = float(np.abs(smp(r["pri"][2]) - g["prior32"]).max())
    head = (r["head"][0] if cfg.mask else r["head"])[0].cpu()
    top = torch.from_numpy(g["head_top_idx"])
    ref_rows = torch.from_numpy(g["head_top_rows"])
    iou = box_iou_pairs(head[top, :4], ref_rows[:, :4])
    met["box_iou_min_top500"] = float(iou.min())
    met["box_iou_mean_top500"] = float(iou.mean())
    sc_ref = ref_rows[:, 4] * ref_rows[:, 5]
    met["score_relerr_top500"] = float(((head[top, 4] * head[top, 5] - sc_ref).abs() / sc_ref).max())
    with torch.no_grad():
        whole, _ = m(frames[1].cuda())
    who = (whole[0] if cfg.mask else whole)[0].cpu()
    wtop, wref = torch.from_numpy(g["whole_top_idx"]), torch.from_numpy(g["whole_top_rows"])
    met["whole_box_iou_min_top500"] = float(box_iou_pairs(who[wtop, :4], wref[:, :4]).min())
    if cfg.mask:
        met["dyn_top64"] = rel_l2(r["head"][2][0].cpu()[top[:64]].numpy(), g["dyn_top_rows"])
        for n, t in (("mask_feats", r["head"][4]), ("up_masks", r["head"][5])):
            met[n] = rel_l2(smp(t), g[n])
        assert np.allclose(smp(r["head"][1]), g["locations"]) and np.array_equal(smp(r["head"][3]), g["fpn_levels"])
        # the reference's postprocess_inst (utils/boxes.py:80-152) on its own head output kept n_det_sot detections; the HIP head + HIP NMS +
        # HIP CondInst must keep the same first ones, with the same masks
        from unicorn_amd.utils.boxes import postprocess_inst
        hd = tuple(t.clone() if torch.is_tensor(t) else t for t in r["head"])
        dets, masks = postprocess_inst(hd[0], hd[1], hd[2], hd[3], hd[4], m.head.mask_head, 1, 0.001, 0.65, d_rate=cfg.d_rate, up_masks=hd[5])
        nref = int(g["n_det_sot"][0])
        met["n_det_sot"] = [int(dets[0].shape[0]), nref]
        dref = torch.from_numpy(g["det_sot"]).reshape(tuple(g["det_sot__shape"]))
        k = dref.shape[0]
        cx = lambda t: torch.stack([(t[:, 0] + t[:, 2]) / 2, (t[:, 1] + t[:, 3]) / 2, t[:, 2] - t[:, 0], t[:, 3] - t[:, 1]], 1)
        met["det_box_iou_min"] = float(box_iou_pairs(cx(dets[0][:k, :4].cpu()), cx(dref[:, :4])).min())
        bits = np.packbits((masks[0][:k].cpu() > 0.5).numpy().astype(np.uint8).reshape(-1))
        diff = np.unpackbits(bits ^ g["mask_sot_bits"]).reshape(k, -1).sum(1)
        area = np.unpackbits(g["mask_sot_bits"]).reshape(k, -1).sum(1)
        met["mask_bits_differ"] = [int(v) for v in diff]
        met["mask_iou_min"] = float(min(1.0 - d / max(1.0, a + d) for d, a in zip(diff, area)))      # union <= area + differing pixels
    METRICS["golden_headline_%s_f16x2_800x1280" % exp] = met
    _dump()
    assert met["seq_pos"] < 1e-5, met
    for k in ("fpn0", "fpn1", "fpn2", "seq_feat", "feat_pre", "feat_cur", "embed_pre", "embed_cur"):
        assert met[k] < 1e-4, (k, met[k])
    assert max(met["coarse_maxabs"], met["prior16_maxabs"], met["prior32_maxabs"]) < 1e-4, met
    assert met["box_iou_min_top500"] > 0.999 and met["whole_box_iou_min_top500"] > 0.999 and met["score_relerr_top500"] < 1e-3, met
    if cfg.mask:
        assert max(met["dyn_top64"], met["mask_feats"], met["up_masks"]) < 1e-4, met
        assert met["n_det_sot"][0] == met["n_det_sot"][1] and met["det_box_iou_min"] > 0.999 and met["mask_iou_min"] > 0.999, met


_ORACLE_CACHE = {}


def _oracle_sot(exp, H, W, P, cfg, frames, box):
    """the CPU oracle costs seconds per frame on the large model: one evaluation per (exp, size) for all precisions"""
    key = (exp, H, W)
    if key not in _ORACLE_CACHE:
        torch.set_num_threads(min(16, os.cpu_count() or 1))   # the GPU box has 100s of cores; torch CPU ops scale badly past ~16
        with torch.no_grad():
            st = uo.sot_init(P, cfg, frames[0], box)
            _ORACLE_CACHE[key] = (st, uo.sot_step(P, cfg, st, frames[1]))
    return _ORACLE_CACHE[key]


def _vs_oracle(exp, H, W, tag, precision="bf16"):
    m, cfg, P = build(exp, precision)
    frames, box = synth.synth_clip(H, W, 2, seed=1)
    r = hip_sot_step(m, cfg, frames, box)
    del m
    st, o = _oracle_sot(exp, H, W, P, cfg, frames, box)
    met = {}
    assert torch.equal(r["lbs"].cpu(), st["lbs_pre"])
    for i in range(3):
        met["fpn%d" % i] = rel_l2(r["fpn"][i].cpu().contiguous().numpy(), o["fpn"][i].numpy())
    met["seq_feat"] = rel_l2(r["seq"]["feat"].cpu().contiguous().numpy(), o["seq"]["feat"].numpy())
    met["feat_cur"] = rel_l2(r["feat_cur"].cpu().contiguous().numpy(), o["feat_cur"].numpy())
    for k in ("embed_pre", "embed_cur"):
        c = cos_per_pixel(r[k], o[k])
        met[k + "_cos_min"] = float(c.min())
        met[k + "_cos_mean"] = float(c.mean())
        met[k + "_rel_l2"] = rel_l2(r[k].cpu().contiguous().numpy(), o[k].numpy())
    met["coarse_maxabs"] = float((r["coarse"].cpu() - o["coarse"]).abs().max())
    # correlation kernel in isolation: HIP corr on the ORACLE's embeddings must match to fp32 round-off
    from unicorn_amd.ops import corr_softmax_pv
    iso = corr_softmax_pv(o["embed_pre"].flatten(-2)[0].cuda(), o["embed_cur"].flatten(-2)[0].cuda(), st["lbs_pre"].cuda())
    met["corr_isolated_maxabs"] = float((iso.cpu() - o["coarse"].flatten(1)).abs().max())
    ho = o["head"][0] if cfg.mask else o["head"]
    hh = (r["head"][0] if cfg.mask else r["head"]).cpu()
    score = ho[0, :, 4] * ho[0, :, 5]
    top = torch.argsort(score, descending=True)[:500]
    iou = box_iou_pairs(hh[0, top, :4], ho[0, top, :4])
    met["box_iou_min_top500"] = float(iou.min())
    met["box_iou_mean_top500"] = float(iou.mean())
    iou_all = box_iou_pairs(hh[0, :, :4], ho[0, :, :4])
    met["box_iou_min_all"] = float(iou_all.min())
    # IoU of a tiny box amplifies fp32 round-off of its corners without bound (a 1 px box moved by 1e-3 px: 0.998), also in the exact-fp32
    # mode (0.9989 on the tiny model): the all-anchor bar is held on boxes of at least one finest-stride cell (8 px) per side
    big = (ho[0, :, 2] >= 8.0) & (ho[0, :, 3] >= 8.0)
    met["box_iou_min_all_8px"] = float(iou_all[big].min()) if bool(big.any()) else 1.0
    met["anchors_8px"] = int(big.sum())
    # the SOT decision (unicorn_sot.py:62-76): NMS, take the first box
    det_o = uo.postprocess(ho.clone(), 1, 0.001, 0.65)[0]
    det_h = uo.postprocess(hh.clone(), 1, 0.001, 0.65)[0]     # same post-processing on both (isolates the network)
    if det_o is not None and det_h is not None:
        a, b = det_h[:1, :4], det_o[:1, :4]
        cx = lambda t: torch.stack([(t[:, 0] + t[:, 2]) / 2, (t[:, 1] + t[:, 3]) / 2, t[:, 2] - t[:, 0], t[:, 3] - t[:, 1]], 1)
        met["sot_box_iou"] = float(box_iou_pairs(cx(a), cx(b))[0])
    if cfg.mask:
        met["dyn_params"] = rel_l2(r["head"][2].cpu().numpy(), o["head"][2].numpy())
        met["mask_feats"] = rel_l2(r["head"][4].cpu().contiguous().numpy(), o["head"][4].numpy())
        met["up_masks"] = rel_l2(r["head"][5].cpu().contiguous().numpy(), o["head"][5].numpy())
        # masks of the oracle's kept detections (utils/boxes.py:80-152): HIP CondInst on HIP head outputs vs oracle
        from unicorn_amd.ops import condinst_masks
        ohead = tuple(t.clone() for t in o["head"])
        det, idx = uo.postprocess(ohead[0], 1, 0.001, 0.65, return_index=True)[0]
        if det is not None:
            idx = idx[:16]
            mo = uo.aligned_bilinear(uo.dynamic_mask_head(cfg, o["head"][4], o["head"][2][0][idx], o["head"][1][idx],
                                                          o["head"][3][0][idx], o["head"][5]), cfg.d_rate)
            mh = condinst_masks(r["head"][4], r["head"][5], r["head"][2][0][idx.cuda()], r["head"][1][idx.cuda()],
                                r["head"][3][0][idx], cfg.up_rate, cfg.d_rate).cpu()
            a, b = mh > 0.5, mo > 0.5
            inter = (a & b).flatten(1).sum(1).float()
            union = (a | b).flatten(1).sum(1).float().clamp_min(1)
            met["mask_iou_min"] = float((inter / union).min())
            met["mask_maxabs"] = float((mh - mo).abs().max())
    METRICS[tag] = met
    _dump()
    return met


def _assert_bar(met, precision):
    feat_tol, prior_tol = (1e-4, 1e-4) if precision in EXACT else (5e-2, 2e-2)
    for k in ("fpn0", "fpn1", "fpn2", "seq_feat", "feat_cur"):
        assert met[k] < feat_tol, (k, met[k])
    assert met["embed_cur_cos_min"] > 1 - 1e-4 and met["embed_pre_cos_min"] > 1 - 1e-4, met     # north_star: cosine within 1e-4
    assert met["corr_isolated_maxabs"] < 2e-5, met
    assert met["coarse_maxabs"] < prior_tol, met
    if precision in EXACT:                                                                   # north_star: IoU >= 0.999
        assert met["box_iou_min_top500"] > 0.999, met
        assert met["box_iou_min_all_8px"] > 0.999, met      # every anchor with a box of at least 8 x 8 px, not only the top-scoring ones
        if "sot_box_iou" in met:
            assert met["sot_box_iou"] > 0.999, met
        if "mask_iou_min" in met:
            assert met["mask_iou_min"] > 0.999, met
    else:
        assert met["box_iou_mean_top500"] > 0.75, met


@pytest.mark.parametrize("precision", ["fp32", "f16x2"])
def test_tiny_sot_800x1280_vs_oracle(precision):
    """BASELINE.json configs[1]: unicorn_track_tiny SOT 800x1280 (bf16 backbone + fp32 correlation; and the exact mode)."""
    _assert_bar(_vs_oracle("unicorn_track_tiny", 800, 1280, "tiny_sot_800x1280_" + precision, precision), precision)


@pytest.mark.parametrize("precision", ["fp32", "f16x2"])
def test_tiny_mask_ragged_size_vs_oracle(precision):
    """VOS-style head (CondInst) at a non-square size that is ragged for every tile/strip/split: 352x608."""
    _assert_bar(_vs_oracle("unicorn_track_tiny_mask", 352, 608, "tiny_mask_352x608_" + precision, precision), precision)


# ------------------------------------------------------------------------------------------------
# the LARGE models at 800x1280: the configurations bench.py times (BASELINE.json configs[2..3] + the headline)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", ["f16x2"])
def test_large_sot_800x1280_vs_oracle(precision):
    """unicorn_track_large SOT step, the bench headline workload: depth-27 stage, C = 1536 tiles, the 256x256 GEMM tiles on
    their real shapes.  f16x2 (the headline precision) must meet the north_star bar; bf16 is held to its documented error
    class (embedding cosine within 1e-4 holds, box IoU does not: profiles/r02_precision_budget.json)."""
    _assert_bar(_vs_oracle("unicorn_track_large", 800, 1280, "large_sot_800x1280_" + precision, precision), precision)


def test_large_mask_sot_and_condinst_800x1280_vs_oracle():
    """unicorn_track_large_mask: head + controllers + mask branch + CondInst masks of the kept detections (config 3's head)."""
    met = _vs_oracle("unicorn_track_large_mask", 800, 1280, "large_mask_800x1280_f16x2", "f16x2")
    _assert_bar(met, "f16x2")
    assert "mask_iou_min" in met and max(met["dyn_params"], met["mask_feats"], met["up_masks"]) < 1e-4, met


def test_large_mot_challenge_evaluate_omni_sequence():
    """BASELINE.json configs[2]: the per-frame sequence of evaluate_omni (unicorn/evaluators/mot_evaluator.py:991-1045) on
    unicorn_track_large_mot_challenge (num_classes = 1): mode="whole" -> postprocess -> interaction(prev, cur) -> ONE upsample ->
    instance embeddings at the box centres -> QuasiDense match, HIP path (f16x2) vs the oracle on 2 frames.  With synthetic
    weights obj*cls ~ 1e-4, so the confidence threshold is set from the oracle's score distribution (same value both sides)."""
    import copy
    import assoc_oracle as ao
    from unicorn_amd.ops import sample_embeddings
    from unicorn_amd.tracker import QuasiDenseEmbedTracker
    from unicorn_amd.utils.boxes import postprocess
    exp, H, W = "unicorn_track_large_mot_challenge", 800, 1280
    m, cfg, P = build(exp, "f16x2")
    frames, _ = synth.synth_clip(H, W, 3, seed=5)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    kw = dict(init_score_thr=0.0, obj_score_thr=0.0, match_score_thr=0.5, memo_tracklet_frames=10, memo_backdrop_frames=1,
              memo_momentum=0.8, nms_conf_thr=0.5, nms_backdrop_iou_thr=0.3, nms_class_iou_thr=0.7, with_cats=True,
              match_metric="bisoftmax")
    trk_h, st_o = QuasiDenseEmbedTracker(**kw), ao.QDState(**kw)
    pre_h = pre_o = None
    thr = None
    met = {"box_iou_min": 1.0, "embed_cos_min": 1.0}
    for fid in (1, 2):
        with torch.no_grad():
            out_o, d_o, _ = uo.mot_whole(P, cfg, frames[fid])
            out_h, d_h = m(frames[fid].cuda())
        assert out_h.shape == out_o.shape == (1, 21000, 6)
        if thr is None:
            sc = (out_o[0, :, 4] * out_o[0, :, 5]).sort(descending=True)[0]
            thr = float((sc[199] + sc[200]) / 2)                        # ~200 candidates before NMS
        det_o = uo.postprocess(out_o.clone(), 1, thr, 0.7)[0]
        det_h = postprocess(out_h.clone(), 1, thr, 0.7)[0]
        assert det_o is not None and det_h is not None and det_h.shape == det_o.shape, (det_h is None, det_o is None)
        # same detection SET: scores of neighbouring rows can differ by less than the fp32-level difference between the two paths,
        # so rows are matched by IoU (must be a bijection) and the HIP rows are put into the oracle's order
        pair = ao.box_iou(det_h[:, :4].cpu(), det_o[:, :4])
        miou, to_o = pair.max(1)
        assert sorted(to_o.tolist()) == list(range(det_o.shape[0])), "detections do not match one-to-one"
        met["box_iou_min"] = min(met["box_iou_min"], float(miou.min()))
        assert miou.min() > 0.999, miou.min()
        det_h = det_h[torch.argsort(to_o).to(det_h.device)]
        assert ((det_h[:, 4] * det_h[:, 5]).cpu() - det_o[:, 4] * det_o[:, 5]).abs().max() < 1e-3 * float((det_o[:, 4] * det_o[:, 5]).max())
        with torch.no_grad():
            if fid == 1:
                pre_h, pre_o = copy.deepcopy(d_h), copy.deepcopy(d_o)    # mot_evaluator.py:1014-1015
            _, f_h = m(seq_dict0=pre_h, seq_dict1=d_h, mode="interaction")
            e_h = m(feat=f_h, mode="upsample")
            _, f_o = uo.forward_interaction(P, pre_o, d_o)
            e_o = uo.forward_upsample(P, f_o)
            pre_h, pre_o = copy.deepcopy(d_h), copy.deepcopy(d_o)
            emb_o = uo.sample_instance_embeddings(e_o, det_o[:, :4])
            emb_h = sample_embeddings(e_h, det_h[:, :4].contiguous()).cpu()
        cos = torch.nn.functional.cosine_similarity(emb_h.double(), emb_o.double(), dim=1)
        met["embed_cos_min"] = min(met["embed_cos_min"], float(cos.min()))
        assert cos.min() > 1 - 1e-4, cos.min()
        labels = torch.ones((det_o.shape[0],))
        tin_o = torch.cat((det_o[:, :4], det_o[:, 4:5] * det_o[:, 5:6]), 1)
        tin_h = torch.cat((det_h[:, :4], det_h[:, 4:5] * det_h[:, 5:6]), 1).cpu()
        b_h, _, ids_h = trk_h.match(tin_h, labels, emb_h, fid)
        b_o, _, ids_o, _ = ao.qd_match(st_o, tin_o, labels, emb_o, fid)
        assert torch.equal(torch.as_tensor(ids_h).long(), torch.as_tensor(ids_o).long()), (ids_h, ids_o)
    METRICS["large_mot_challenge_omni_f16x2"] = met
    _dump()


def test_large_mask_vos_k3_tracker_step_vs_oracle():
    """BASELINE.json configs[3]: unicorn_track_large_mask VOS step with K = 3 objects (UnicornVOSTrack: correlation, head,
    postprocess_inst, CondInst mask of the best instance) vs the oracle's per-object loop, f16x2."""
    from unicorn_amd.tracker import UnicornVOSTrack
    m, cfg, P = build("unicorn_track_large_mask", "f16x2")
    H, W = 800, 1280
    frames, box = synth.synth_clip(H, W, 2, seed=3)
    boxes = {"1": box, "2": torch.tensor([W * 0.55, H * 0.1, W * 0.9, H * 0.45]), "3": torch.tensor([W * 0.1, H * 0.55, W * 0.4, H * 0.95])}
    trk = UnicornVOSTrack(m, input_size=(H, W), d_rate=cfg.d_rate)
    trk.initialize(frames[0].cuda(), {"init_object_ids": list(boxes), "init_bbox": {k: [float(b[0]), float(b[1]), float(b[2] - b[0]), float(b[3] - b[1])]
                                                                                    for k, b in boxes.items()}})
    res, _ = trk.step(frames[1].cuda())
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        st = uo.vos_init(P, cfg, frames[0], boxes)
        exp = uo.vos_step(P, cfg, st, frames[1])
    met = {"mask_iou_min": 1.0, "box_iou_min": 1.0}
    for k in boxes:
        d_o, m_o = exp[k]
        d_h, m_h = res[k]
        assert (d_o is None) == (d_h is None), k
        if d_o is None:
            continue
        cx = lambda t: torch.stack([(t[0] + t[2]) / 2, (t[1] + t[3]) / 2, t[2] - t[0], t[3] - t[1]])[None]
        met["box_iou_min"] = min(met["box_iou_min"], float(box_iou_pairs(cx(d_h.cpu()), cx(d_o))[0]))
        a, b = m_h.cpu() > 0.5, m_o > 0.5
        met["mask_iou_min"] = min(met["mask_iou_min"], float((a & b).sum()) / max(float((a | b).sum()), 1.0))
    METRICS["large_mask_vos_k3_f16x2"] = met
    _dump()
    assert met["box_iou_min"] > 0.999 and met["mask_iou_min"] > 0.999, met


@pytest.mark.parametrize("exp", ["unicorn_track_tiny_mask", "unicorn_track_large_mask"])
def test_vos_tracker_step_vs_reference_golden(exp, golden_dir):
    """UnicornVOSTrack.step (K = 3 objects, one correlation pass + one object-batched head call) against vectors of the REAL reference
    (tests/golden/make_golden.py:run_vos = unicorn_vos.py:157-200 on the reference model): best box per object, mask bits, f16x2."""
    from unicorn_amd.tracker import UnicornVOSTrack
    H = W = 320
    g = np.load(os.path.join(golden_dir, "%s_vos_%dx%d.npz" % (exp, H, W)))
    m, cfg, P = build(exp, "f16x2")
    frames, box = synth.synth_clip(H, W, 2, seed=1)
    boxes = {"1": box, "2": torch.tensor([W * 0.55, H * 0.1, W * 0.9, H * 0.45]), "3": torch.tensor([W * 0.1, H * 0.55, W * 0.4, H * 0.95])}
    trk = UnicornVOSTrack(m, input_size=(H, W), d_rate=cfg.d_rate)
    trk.initialize(frames[0].cuda(), {"init_object_ids": list(boxes), "init_bbox": {k: [float(b[0]), float(b[1]), float(b[2] - b[0]), float(b[3] - b[1])]
                                                                                    for k, b in boxes.items()}})
    res, _ = trk.step(frames[1].cuda())
    cx = lambda t: torch.stack([(t[0] + t[2]) / 2, (t[1] + t[3]) / 2, t[2] - t[0], t[3] - t[1]])[None]
    for k in boxes:
        d_h, m_h = res[k]
        assert (d_h is None) == (int(g["n_det_%s" % k][0]) == 0), k
        if d_h is None:
            continue
        ref = torch.from_numpy(g["det_%s" % k])
        ref[0:4:2] = ref[0:4:2].clamp(0, W)                          # the driver clamps to the input size (unicorn_vos.py:133-134)
        ref[1:4:2] = ref[1:4:2].clamp(0, H)
        assert float(box_iou_pairs(cx(d_h.cpu()), cx(ref))[0]) > 0.999, (k, d_h, ref)
        bits = np.packbits((m_h.cpu() > 0.5).numpy().astype(np.uint8).reshape(-1))
        diff = np.unpackbits(bits ^ g["mask_bits_%s" % k]).sum()
        assert diff <= 1e-3 * m_h.numel(), (k, int(diff))


def test_vos_tracker_reference_groups_raw_images_vs_oracle():
    """Row N3a + N1: the full driver on RAW uint8 frames (letterbox r != 1, so the resize of the aggregation is exercised),
    with an object that APPEARS AT FRAME 2 (info["init_object_ids"] / init_bbox / init_mask, unicorn_vos.py:87-98): from frame 3
    on it is tracked against its own reference frame.  HIP tracker vs oracle.vos_track_frame, frame by frame."""
    import letterbox_oracle as lo
    from unicorn_amd.tracker import UnicornVOSTrack
    m, cfg, P = build("unicorn_track_tiny_mask", "f16x2")
    h, w, size = 240, 400, (320, 512)
    g = np.random.default_rng(7)
    base = g.integers(0, 256, (h, w, 3), dtype=np.uint8)
    imgs = [np.roll(base, (2 * t, 3 * t), (0, 1)) for t in range(4)]
    b0 = {"1": [60.0, 40.0, 120.0, 90.0], "2": [230.0, 110.0, 100.0, 80.0]}              # xywh on the original image
    new = {"3": [20.0, 150.0, 90.0, 70.0]}
    init_mask = np.zeros((h, w), dtype=np.uint8)
    init_mask[150:220, 20:110] = 3
    trk = UnicornVOSTrack(m, input_size=size, d_rate=cfg.d_rate)
    trk.initialize(imgs[0], {"init_object_ids": list(b0), "init_bbox": b0})
    xyxy = lambda b: torch.tensor([b[0], b[1], b[0] + b[2], b[1] + b[3]])
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    t0, r = lo.letterbox(imgs[0], size, True)
    with torch.no_grad():
        st = uo.vos_track_init(P, cfg, torch.from_numpy(t0)[None], {k: xyxy(b) for k, b in b0.items()}, (h, w), r)
    agree = []
    for t in (1, 2, 3):
        info = {"init_object_ids": list(new), "init_bbox": new, "init_mask": init_mask} if t == 2 else {}
        seg = trk.track(imgs[t], dict(info))["segmentation"]
        tt, r = lo.letterbox(imgs[t], size, True)
        oinfo = {"init_object_ids": list(new), "init_bbox": {k: xyxy(b) for k, b in new.items()}, "init_mask": init_mask} if t == 2 else {}
        with torch.no_grad():
            exp = uo.vos_track_frame(P, cfg, st, torch.from_numpy(tt)[None], oinfo, r)
        assert seg.shape == exp.shape == (h, w) and seg.dtype == np.uint8
        agree.append(float((seg == exp).mean()))
        if t == 2:
            assert (seg[init_mask == 3] == 3).mean() > 0.9             # the given mask of the new object (ties with a saturated p = 1 of a lower id aside)
    METRICS["vos_reference_groups_agreement"] = agree
    _dump()
    assert len(trk.obj_ids_new) == 1 and trk.obj_ids_new[0] == ["3"] and len(st["groups"]) == 2
    assert min(agree) > 0.999, agree                                   # id maps agree (fp32-grade masks; ties at mask borders aside)


def test_whole_mot_mode_matches_head_with_zero_priors():
    m, cfg, P = build("unicorn_track_tiny", "fp32")
    frames, _ = synth.synth_clip(320, 320, 2, seed=1)
    with torch.no_grad():
        out, seq = m(frames[1].cuda())
        whole_o, _, _ = uo.mot_whole(P, cfg, frames[1])
    assert out.shape == (1, 2100, 13) and seq["feat"].shape == (1, 384, 20, 20)
    score = whole_o[0, :, 4] * whole_o[0, :, 5:].max(1)[0]
    top = torch.argsort(score, descending=True)[:200]
    iou = box_iou_pairs(out[0].cpu()[top, :4], whole_o[0, top, :4])
    assert iou.min() > 0.999
    with pytest.raises(ValueError):
        m.head(None, None, mode="bogus")
    with pytest.raises(ValueError):
        m(imgs=frames[1].cuda(), mode="nonsense")


def test_sot_tracker_and_postprocess_match_oracle_decision():
    """UnicornSOTTrack (mirror of external/lib/test/tracker/unicorn_sot.py) on the exact-fp32 mode: the integer box
    it reports must equal the oracle's decision (postprocess conf 0.001 / nms 0.65, clamp, index 0, int truncation)."""
    from unicorn_amd.tracker import UnicornSOTTrack
    from unicorn_amd.utils.boxes import postprocess
    m, cfg, P = build("unicorn_track_tiny", "fp32")
    H = W = 320
    frames, box = synth.synth_clip(H, W, 3, seed=1)
    trk = UnicornSOTTrack(m, input_size=(H, W))
    xywh = [float(box[0]), float(box[1]), float(box[2] - box[0]), float(box[3] - box[1])]
    trk.initialize(frames[0].cuda(), {"init_bbox": xywh})
    got = [trk.track(frames[i].cuda())["target_bbox"] for i in (1, 2)]
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        st = uo.sot_init(P, cfg, frames[0], box)
        exp = []
        for i in (1, 2):
            o = uo.sot_step(P, cfg, st, frames[i])
            det = uo.postprocess(o["head"].clone(), 1, 0.001, 0.65)[0]
            exp.append(uo.sot_pick_box(det, H, W))
            if i == 1:   # HIP-side postprocess on the oracle's head output: same kept rows, same order
                hd = postprocess(o["head"].clone().cuda(), 1, 0.001, 0.65)[0]
                assert hd.shape == det.shape and torch.allclose(hd.cpu(), det, atol=1e-4)
    for a, b in zip(got, exp):
        assert all(abs(x - y) <= 1 for x, y in zip(a, b)), (got, exp)     # int truncation may flip by 1 px


@pytest.mark.parametrize("precision", ["fp32", "f16x2", "bf16"])      # bf16 (retired from the bench, still a public mode): ONE batched + ragged case
def test_batched_frames_equal_single_frame_runs(precision):
    """Every stage takes a batch of frames (M = B*H*W rows per kernel): B = 3 must reproduce three B = 1 runs
    (per-sample GroupNorm statistics, conv halos and the head's row remap must not leak across samples).
    Ragged 352x608: none of the per-level pixel counts is a multiple of the GEMM tiles, so tiles straddle samples."""
    from unicorn_amd.ops import corr_softmax_pv, label_map_s8, prior_pyramid
    m, cfg, P = build("unicorn_track_tiny_mask", precision)
    H, W = 352, 608
    frames, box = synth.synth_clip(H, W, 4, seed=3)
    ref = frames[0].cuda()
    cur = torch.cat(frames[1:4], 0).cuda()
    with torch.no_grad():
        _, d_pre = m(imgs=ref, mode="backbone")
        lbs = label_map_s8(box, H, W, "cuda")

        def run(x):
            fpn, d = m(imgs=x, mode="backbone")
            fp, fc = m(seq_dict0=d_pre, seq_dict1=d, mode="interaction")
            ep, ec = m(feat=fp, mode="upsample"), m(feat=fc, mode="upsample")
            pri = torch.cat([corr_softmax_pv(ep[b].flatten(-2), ec[b].flatten(-2), lbs).view(1, 1, d["h"] * 2, d["w"] * 2)
                             for b in range(x.shape[0])], 0)
            pp = tuple(t.transpose(0, 1).contiguous() for t in prior_pyramid(pri.transpose(0, 1).contiguous()))
            out = m.head(fpn, pp, mode="sot")
            return fpn, d["feat"], fc, ec, pri, out
        big = run(cur)
        singles = [run(cur[b:b + 1]) for b in range(3)]
    torch.cuda.synchronize()
    # bf16: the fp64 GN-statistics atomics arrive in a different order -> a different last bit of rstd can flip a bf16
    # rounding, which the random-weight head then amplifies; the exact-fp32 mode pins the batching logic itself
    tol = 5e-5 if precision in EXACT else 1e-1      # (B = 1 and B = 3 take different GEMM tile shapes and fp64-atomic orders: 1.0e-5 .. 2.3e-5 measured over boxes / rounds)
    def close(a, b, what):
        a, b = a.float().cpu(), b.float().cpu()
        err = (a - b).abs().max().item() / max(1.0, b.abs().max().item())
        assert err < tol, (what, err)
    for b in range(3):
        for k in range(3):
            close(big[0][k][b:b + 1], singles[b][0][k], "fpn%d" % k)
        close(big[1][b:b + 1], singles[b][1], "feat16")
        close(big[2][b:b + 1], singles[b][2], "feat_cur")
        close(big[3][b:b + 1], singles[b][3], "embed_cur")
        close(big[4][b:b + 1], singles[b][4], "prior")
        for j in (0, 2, 4, 5):     # outputs, dynamic params, mask feats, up masks
            close(big[5][j][b:b + 1], singles[b][5][j], "head[%d]" % j)


@pytest.mark.parametrize("batched", [True, False])
def test_vos_tracker_object_batched_matches_oracle(batched):
    """UnicornVOSTrack (mirror of external/lib/test/tracker/unicorn_vos.py, row N3): ONE correlation + ONE batched head call
    for all objects must reproduce the oracle's per-object loop (exact-fp32 mode): best box, mask, merged segmentation."""
    from unicorn_amd.tracker import UnicornVOSTrack
    m, cfg, P = build("unicorn_track_tiny_mask", "fp32")
    H = W = 320
    frames, box = synth.synth_clip(H, W, 2, seed=3)
    boxes = {"1": box, "2": torch.tensor([40.0, 60.0, 150.0, 170.0]), "3": torch.tensor([180.0, 30.0, 300.0, 140.0])}
    trk = UnicornVOSTrack(m, input_size=(H, W), d_rate=cfg.d_rate, object_batched=batched)
    info = {"init_object_ids": list(boxes), "init_bbox": {k: [float(b[0]), float(b[1]), float(b[2] - b[0]), float(b[3] - b[1])]
                                                          for k, b in boxes.items()}}
    trk.initialize(frames[0].cuda(), info)
    res, r = trk.step(frames[1].cuda())
    seg = trk.track(frames[1].cuda())["segmentation"]
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        st = uo.vos_init(P, cfg, frames[0], boxes)
        exp = uo.vos_step(P, cfg, st, frames[1])
    prob = {}
    for k in boxes:
        d_o, m_o = exp[k]
        d_h, m_h = res[k]
        assert (d_o is None) == (d_h is None), k
        if d_o is None:
            prob[k] = np.zeros((H, W), np.float32)
            continue
        cx = lambda t: torch.stack([(t[0] + t[2]) / 2, (t[1] + t[3]) / 2, t[2] - t[0], t[3] - t[1]])[None]
        assert float(box_iou_pairs(cx(d_h.cpu()), cx(d_o))[0]) > 0.999, k
        assert abs(float(d_h[4] * d_h[5]) - float(d_o[4] * d_o[5])) < 1e-5
        a, b = m_h.cpu() > 0.5, m_o > 0.5
        assert float((a & b).sum()) / max(float((a | b).sum()), 1.0) > 0.995, k
        assert float((m_h.cpu() - m_o).abs().max()) < 2e-3, k
        prob[k] = m_o.numpy()
    assert (seg == uo.vos_merge(prob, H, W)).mean() > 0.999


def test_load_state_dict_reports_missing_and_unexpected():
    """nn.Module.load_state_dict contract used by tools/track.py:188 (`strict=False` -> (missing, unexpected))"""
    from unicorn_amd.models import Unicorn
    cfg = uo.CONFIGS["unicorn_track_tiny"]
    P = dict(synth.synth_state_dict(cfg))
    del P["head.beta_1"]
    P["head.some_new_tensor"] = torch.zeros(3)
    m = Unicorn("unicorn_track_tiny").cuda(0)
    res = m.load_state_dict(P, strict=False)
    assert list(res.missing_keys) == ["head.beta_1"] and list(res.unexpected_keys) == ["head.some_new_tensor"]
    m2 = Unicorn("unicorn_track_tiny").cuda(0)
    with pytest.raises(RuntimeError):
        m2.load_state_dict(P, strict=True)
    # reference order load -> cuda (tools/track.py:176-188): same report before the device exists, strict honoured,
    # and a wrong-config checkpoint (nc = 1 into nc = 8) raises instead of running on zero-filled weights
    m3 = Unicorn("unicorn_track_tiny")
    res = m3.load_state_dict(P, strict=False)
    assert list(res.missing_keys) == ["head.beta_1"] and list(res.unexpected_keys) == ["head.some_new_tensor"]
    with pytest.raises(RuntimeError):
        Unicorn("unicorn_track_tiny").load_state_dict(P, strict=True)
    Pw = dict(synth.synth_state_dict(uo.CONFIGS["unicorn_track_tiny"]))
    Pw["head.cls_preds.0.weight"] = Pw["head.cls_preds.0.weight"][:1]
    with pytest.raises(RuntimeError, match="size mismatch"):
        Unicorn("unicorn_track_tiny").load_state_dict(Pw, strict=False)
    m3.cuda(0)
    frames, _ = synth.synth_clip(64, 64, 1, seed=0)
    fpn, _ = m3(imgs=frames[0].cuda(), mode="backbone")
    assert torch.isfinite(fpn[0]).all()


def test_raw_outputs_decode_in_inference_false():
    """tools/export_torchscript.py:66 sets model.head.decode_in_inference = False: the head then returns the undecoded rows
    [reg(4), sigmoid(obj), sigmoid(cls)] (unicorn_head.py:430-439); head.decode_outputs on them equals the decoded output
    (tools/track.py:208-209 `decoder`).  The mask head has no such path (unicorn_head_mask.py:470 raises ValueError)."""
    m, cfg, P = build("unicorn_track_tiny", "f16x2")
    frames, _ = synth.synth_clip(320, 320, 2, seed=1)
    with torch.no_grad():
        fpn, _ = m(imgs=frames[1].cuda(), mode="backbone")
        pri = tuple(torch.zeros((1, 1, 320 // s, 320 // s), device="cuda") for s in (8, 16, 32))
        dec = m.head(fpn, pri, mode="mot").clone()
        m.head.decode_in_inference = False
        raw = m.head(fpn, pri, mode="mot")
        m.head.decode_in_inference = True
        fo, _ = uo.forward_backbone(P, cfg, frames[1])
        outs, _ = uo._head_trunk(P, cfg, fo, tuple(p.cpu() for p in pri), "mot")
    raw_o = torch.cat([o.flatten(2) for o in outs], 2).permute(0, 2, 1)
    assert raw.shape == raw_o.shape == (1, 2100, 13)
    assert (raw.cpu() - raw_o).abs().max() < 2e-4
    assert torch.allclose(m.head.decode_outputs(raw.clone()), dec, atol=1e-5, rtol=1e-6)
    mm, _, _ = build("unicorn_track_tiny_mask", "f16x2")
    mm.head.decode_in_inference = False
    with pytest.raises(ValueError):
        mm.head(fpn, pri, mode="mot")


@pytest.mark.parametrize("env", [{"UNI_MLP_LAYOUT": "0"}, {"UNI_NO_MLP_FUSED": "1", "UNI_NO_SPLITK": "1"}, {"UNI_NO_H2D": "1"}],
                         ids=["mlp32", "unfused_nosplit", "no_deep_tiles"])
def test_engine_ab_switches_keep_parity(env):
    """The launch-plan switches of the engine (read when the weights are packed / per call) select kernels that are no longer the
    default: the 32-row fused MLP (UNI_MLP_LAYOUT=0), the two-launch MLP and unsplit single-frame convolutions, the generic two-stage
    tiles instead of the deep-pipeline ones (UNI_NO_H2D=1).  The large SOT parity
    test must pass with each of them."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k", "test_large_sot_800x1280_vs_oracle and f16x2"],
                       env=dict(os.environ, **env), capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0 and "1 passed" in r.stdout, r.stdout[-1500:] + r.stderr[-500:]


def test_omni_mot_frame_batched_equals_per_frame_and_oracle():
    """unicorn_amd.tracker.OmniMOTFrame = the loop body of MOTEvaluator.evaluate_omni (mot_evaluator.py:991-1045).  Three frames of the
    tiny model: (a) one frame per call vs all frames in ONE time-batched call give the same boxes and ids, (b) the ids / boxes are
    the oracle's (uo.mot_whole -> uo.postprocess -> interaction / upsample -> sample_instance_embeddings -> assoc_oracle.qd_match)."""
    import copy
    import assoc_oracle as ao
    from unicorn_amd.tracker import OmniMOTFrame, QuasiDenseEmbedTracker
    exp, H, W = "unicorn_track_tiny", 320, 320
    m, cfg, P = build(exp, "f16x2")
    frames, _ = synth.synth_clip(H, W, 4, seed=7)
    kw = dict(init_score_thr=0.0, obj_score_thr=0.0, match_score_thr=0.5, memo_tracklet_frames=10, memo_backdrop_frames=1,
              memo_momentum=0.8, nms_conf_thr=0.5, nms_backdrop_iou_thr=0.3, nms_class_iou_thr=0.7, with_cats=True,
              match_metric="bisoftmax")
    with torch.no_grad():
        out_o, _, _ = uo.mot_whole(P, cfg, frames[1])
    sc = (out_o[0, :, 4] * out_o[0, :, 5:].max(1)[0]).sort(descending=True)[0]
    thr = float((sc[59] + sc[60]) / 2)
    info = (480, 480)                                                 # original image: scale = 320 / 480
    one = OmniMOTFrame(m, QuasiDenseEmbedTracker(**kw), (H, W), num_classes=cfg.num_classes, confthre=thr, nmsthre=0.7, embed_score_thr=thr)
    bat = OmniMOTFrame(m, QuasiDenseEmbedTracker(**kw), (H, W), num_classes=cfg.num_classes, confthre=thr, nmsthre=0.7, embed_score_thr=thr)
    with torch.no_grad():
        r1 = [one.run(frames[f].cuda(), info) for f in (1, 2, 3)]
        rb = bat.run_batch(torch.cat([frames[f] for f in (1, 2, 3)], 0).cuda(), info)
    for (b1, i1), (b2, i2) in zip(r1, rb):
        assert torch.equal(torch.as_tensor(i1), torch.as_tensor(i2))
        assert (torch.as_tensor(b1) - torch.as_tensor(b2)).abs().max() < 1e-2      # boxes of ~400 px; B = 1 and B = 3 take different tile shapes
    # oracle loop
    st, pre = ao.QDState(**kw), None
    scale = min(H / 480.0, W / 480.0)
    for k, f in enumerate((1, 2, 3)):
        with torch.no_grad():
            o, d, _ = uo.mot_whole(P, cfg, frames[f])
            det = uo.postprocess(o.clone(), cfg.num_classes, thr, 0.7)[0]
            if pre is None:
                pre = copy.deepcopy(d)
            _, fo = uo.forward_interaction(P, pre, d)
            e = uo.forward_upsample(P, fo)
            pre = copy.deepcopy(d)
        bb, s_ = det[:, :4], det[:, 4:5] * det[:, 5:6]
        keep = s_[:, 0] > thr
        bb, s_ = bb[keep], s_[keep]
        emb = uo.sample_instance_embeddings(e, bb)
        tin = torch.cat((bb / scale, s_), 1)
        b_o, _, ids_o, _ = ao.qd_match(st, tin, torch.ones((bb.shape[0],)), emb, k + 1)
        ids_o = torch.as_tensor(ids_o).long()
        v = ids_o > -1
        b_h, i_h = r1[k]
        # scores of neighbouring detections can swap rows between the two paths: compare as id -> box maps
        mo_ = {int(i): torch.as_tensor(b_o)[v][j] for j, i in enumerate(ids_o[v].tolist())}
        mh_ = {int(i): torch.as_tensor(b_h)[j] for j, i in enumerate(torch.as_tensor(i_h).tolist())}
        assert len(mo_) == len(mh_) and len(mo_) > 10
        hit = 0
        for bo in mo_.values():
            d_ = torch.stack([(bo[:4] - bh[:4]).abs().max() for bh in mh_.values()])
            hit += int(d_.min() < 0.05)
        assert hit == len(mo_), (hit, len(mo_))


@pytest.mark.parametrize("exp,H,W,nframes", [("unicorn_track_tiny", 320, 320, 5), ("unicorn_track_large_mot_challenge", 800, 1280, 2)])
def test_byte_mot_frame_tools_track_loop_vs_oracle(exp, H, W, nframes):
    """`tools/track.py`'s own per-frame loop = MOTEvaluator.evaluate (unicorn/evaluators/mot_evaluator.py:198-222): model(imgs) (mode="whole") ->
    postprocess -> BYTETracker.update -> area / aspect filter, on the HIP path (`unicorn_amd.tracker.ByteMOTFrame` + the native BYTETracker)
    against the oracle loop (uo.mot_whole -> uo.postprocess -> oracle/bytetrack_oracle.byte_update, itself pinned by golden sequences of the REAL
    reference class).  Detector-like scores come from planted obj / cls biases (tests/planted.py:confident_head); thresholds are placed
    between two neighbouring oracle scores so that a 1e-6 score difference cannot flip a decision.  Per frame: same track ids, boxes within
    0.05 px, scores within 1e-5; `run_stream` (pipelined) gives exactly the per-frame results."""
    from types import SimpleNamespace
    import bytetrack_oracle as bo
    from planted import confident_head
    from unicorn_amd.models import Unicorn
    from unicorn_amd.tracker import BYTETracker, ByteMOTFrame
    from unicorn_amd.tracker import byte_tracker as bt
    cfg = uo.CONFIGS[exp]
    # score distributions wide enough for det_thresh = track_thresh + 0.1 to leave new tracks: tiny 0.32 - 0.93 with the default planting; the large
    # head's logits vary less (std ~0.5), so its obj / cls prediction weights are doubled around zero biases (top-300 scores 0.53 - 0.94)
    big = exp != "unicorn_track_tiny"
    P = confident_head(synth.synth_state_dict(cfg), 0.0, 0.0, 2.0) if big else confident_head(synth.synth_state_dict(cfg))
    m = Unicorn(exp, precision="f16x2").cuda()
    assert not m.load_state_dict(P, strict=False)[0]
    m.eval()
    frames, _ = synth.synth_clip(H, W, nframes + 1, seed=7)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        outs_o = [uo.mot_whole(P, cfg, frames[f])[0] for f in range(1, nframes + 1)]
    sc = (outs_o[0][0, :, 4] * outs_o[0][0, :, 5:5 + cfg.num_classes].max(1)[0]).sort(descending=True)[0]
    nconf, ntrk = (299, 149) if big else (149, 59)
    confthre = float((sc[nconf] + sc[nconf + 1]) / 2)
    track_thresh = float((sc[ntrk] + sc[ntrk + 1]) / 2)
    info = (int(H * 1.5), int(W * 1.5), 1, 1, "seq/000001.jpg")                      # info_imgs: (img_h, img_w, frame_id, video_id, file_name)
    args = SimpleNamespace(track_thresh=track_thresh, track_buffer=30, match_thresh=0.9, mot20=False)      # tools/track.py:104-108
    bt.clean_id()
    one = ByteMOTFrame(m, BYTETracker(args), (H, W), num_classes=cfg.num_classes, confthre=confthre, nmsthre=0.7, min_box_area=10)
    with torch.no_grad():
        res_h = [one.run(frames[f].cuda(), info) for f in range(1, nframes + 1)]
    st = bo.ByteState(track_thresh=track_thresh, track_buffer=30, match_thresh=0.9, mot20=False, frame_rate=30)
    n_tracks = 0
    for k in range(nframes):
        det = uo.postprocess(outs_o[k].clone(), cfg.num_classes, confthre, 0.7)[0]
        assert det is not None and res_h[k] is not None
        tracks = bo.byte_update(st, det.numpy(), info, (H, W))
        exp_rows = {}
        for t in tracks:
            tlwh = np.asarray(t.tlwh, dtype=np.float64)
            if tlwh[2] * tlwh[3] > 10 and not (tlwh[2] / tlwh[3] > 1.6):      # (min_box_area 10: the synthetic heads regress small boxes)
                exp_rows[int(t.track_id)] = (tlwh, float(t.score))
        tl_h, ids_h, sc_h = res_h[k]
        got = {int(i): (np.asarray(b, dtype=np.float64), float(s)) for b, i, s in zip(tl_h, ids_h, sc_h)}
        # ids are handed out in detection order: a swap of two near-equal scores would permute them -- compare as box sets first
        assert len(got) == len(exp_rows), (k, len(got), len(exp_rows))
        for tid, (b, s_) in exp_rows.items():
            d = min(float(np.abs(b - g_[0]).max()) for g_ in got.values())
            assert d < 0.05, (k, tid, d)
        same_ids = sorted(got) == sorted(exp_rows)
        if same_ids:
            for tid, (b, s_) in exp_rows.items():
                assert float(np.abs(b - got[tid][0]).max()) < 0.05 and abs(s_ - got[tid][1]) < 1e-5, (k, tid)
        METRICS.setdefault("byte_loop_%s" % exp, []).append({"frame": k + 1, "tracks": len(got), "ids_identical": bool(same_ids)})
        assert same_ids, (k, sorted(got)[:10], sorted(exp_rows)[:10])
        n_tracks = max(n_tracks, len(got))
    assert n_tracks >= 5, n_tracks
    # the software-pipelined stream gives the per-frame results
    bt.clean_id()
    two = ByteMOTFrame(m, BYTETracker(args), (H, W), num_classes=cfg.num_classes, confthre=confthre, nmsthre=0.7, min_box_area=10)
    with torch.no_grad():
        res_s = [r[0] for r in two.run_stream((frames[f].cuda() for f in range(1, nframes + 1)), info)]
    for a, b in zip(res_h, res_s):      # (the GroupNorm sums are fp64 atomics: two runs of the network may differ in the last bit of a box)
        assert list(a[1]) == list(b[1]) and all(float(np.abs(x - y).max()) < 1e-3 for x, y in zip(a[0], b[0]))
    _dump()


def test_demo_predictor_class_agnostic_vs_oracle():
    """tools/demo.py:136-172 (`Predictor.inference`, detection / tracking experiments): raw HWC uint8 BGR image -> ValTransform letterbox (no channel
    swap, pad 114) -> model(img) (mode="whole") -> postprocess(..., class_agnostic=True) -- `unicorn_amd.tracker.DemoPredictor` against
    oracle/letterbox_oracle.py + uo.mot_whole + uo.postprocess(class_agnostic=True) on the 8-class BDD head (class-agnostic NMS suppresses
    ACROSS classes, which the per-class path of the other loops never does)."""
    import letterbox_oracle as lo
    from planted import confident_head
    from unicorn_amd.models import Unicorn
    from unicorn_amd.tracker import DemoPredictor
    exp, size = "unicorn_track_tiny", (320, 512)
    cfg = uo.CONFIGS[exp]
    P = confident_head(synth.synth_state_dict(cfg))
    m = Unicorn(exp, precision="f16x2").cuda()
    assert not m.load_state_dict(P, strict=False)[0]
    m.eval()
    img = np.random.default_rng(11).integers(0, 256, (270, 400, 3), dtype=np.uint8)
    t, r = lo.letterbox(img, size, False)
    with torch.no_grad():
        o, _, _ = uo.mot_whole(P, cfg, torch.from_numpy(t)[None])
    sc = (o[0, :, 4] * o[0, :, 5:5 + cfg.num_classes].max(1)[0]).sort(descending=True)[0]
    confthre = float((sc[299] + sc[300]) / 2)
    det_o = uo.postprocess(o.clone(), cfg.num_classes, confthre, 0.45, class_agnostic=True)[0]
    det_pc = uo.postprocess(o.clone(), cfg.num_classes, confthre, 0.45, class_agnostic=False)[0]
    pred = DemoPredictor(m, cfg.num_classes, confthre, 0.45, size)
    dets, info = pred.inference(img)
    det_h = dets[0].cpu()
    assert abs(info["ratio"] - r) < 1e-12 and info["height"] == 270 and info["width"] == 400
    assert det_o.shape[0] < det_pc.shape[0]                                          # the agnostic NMS really suppresses across classes here
    assert det_h.shape == det_o.shape, (det_h.shape, det_o.shape)
    assert torch.equal(det_h[:, 6], det_o[:, 6])                                      # same survivors in the same (score) order
    assert (det_h[:, :4] - det_o[:, :4]).abs().max() < 0.05 and (det_h[:, 4:6] - det_o[:, 4:6]).abs().max() < 1e-5


def test_omni_stream_pipelined_equals_per_frame():
    """run_stream (software pipeline over the launch stream: A0 A1 B0 A2 | H0 C0 B1 A3 | ...) must give exactly the per-frame results:
    MOT loop (ids, boxes) and MOTS loop (1-based ids, RLE strings), and the SOT driver's track_stream the boxes of track()."""
    from unicorn_amd.tracker import OmniMOTFrame, OmniMOTSFrame, QuasiDenseEmbedTracker, UnicornSOTTrack
    H = W = 320
    kw = dict(init_score_thr=0.0, obj_score_thr=0.0, match_score_thr=0.5, memo_tracklet_frames=10, memo_backdrop_frames=1,
              memo_momentum=0.8, nms_conf_thr=0.5, nms_backdrop_iou_thr=0.3, nms_class_iou_thr=0.7, with_cats=True,
              match_metric="bisoftmax")
    frames, box = synth.synth_clip(H, W, 7, seed=7)
    fr = [frames[f].cuda() for f in range(1, 7)]
    info = (480, 480)
    # ---- MOT
    m, cfg, P = build("unicorn_track_tiny", "f16x2")
    with torch.no_grad():
        o, _ = m(fr[0])
    sc = (o[0, :, 4] * o[0, :, 5:].max(1)[0]).sort(descending=True)[0]
    thr = float((sc[59] + sc[60]) / 2)
    mk = lambda: OmniMOTFrame(m, QuasiDenseEmbedTracker(**kw), (H, W), num_classes=cfg.num_classes, confthre=thr, nmsthre=0.7, embed_score_thr=thr)
    one, pip = mk(), mk()
    with torch.no_grad():
        r1 = [one.run(f, info) for f in fr]
        rp = [r[0] for r in pip.run_stream(fr, info)]
    assert len(rp) == len(r1) == 6
    for (b1, i1), (b2, i2) in zip(r1, rp):
        # same kernels, same order of work per frame; GroupNorm sums are accumulated with fp64 atomics (order-dependent in the last bits)
        assert torch.equal(i1, i2) and (b1 - b2).abs().max() < 1e-2 and len(i1) > 10
        assert torch.equal(i1, i1.sort()[0])                          # ascending track ids (mot_evaluator.py:1052-1055)
    # a high threshold on the score empties some frames: the loop must survive frames without detections (result (None, None))
    hi = mk()
    hi.confthre = 0.99                                               # synthetic scores are << 1: no frame has a detection
    with torch.no_grad():
        assert all(r[0] == (None, None) for r in hi.run_stream(fr[:3], info)) and hi.pre_dict is None and hi.frame_id == 3
    del m
    # ---- MOTS
    mm, cfgm, _ = build("unicorn_track_tiny_mask", "f16x2")
    with torch.no_grad():
        o, _ = mm(fr[0])
    sc = (o[0][0, :, 4] * o[0][0, :, 5:].max(1)[0]).sort(descending=True)[0]
    thr = float((sc[19] + sc[20]) / 2)
    mks = lambda: OmniMOTSFrame(mm, QuasiDenseEmbedTracker(**kw), (H, W), num_classes=cfgm.num_classes, confthre=thr, nmsthre=0.7,
                                embed_score_thr=thr, mask_thres=0.3, d_rate=cfgm.d_rate, min_box_area=10)
    one, pip = mks(), mks()
    with torch.no_grad():
        r1 = [one.run(f, info) for f in fr]
        rp = [r[0] for r in pip.run_stream(fr, info)]
    assert [r[0] for r in r1] == [r[0] for r in rp]
    same = sum(int(a == b) for x, y in zip(r1, rp) for a, b in zip(x[1], y[1]))
    assert same >= 0.9 * sum(len(r[1]) for r in r1), same       # RLE strings: identical unless a probability sits within round-off of the threshold
    assert sum(len(r[1]) for r in r1) > 20 and all(isinstance(s_, str) for r in r1 for s_ in r[1])
    # the fused CondInst -> resized bytes entry point (uni_condinst_masks_u8, the default) against the two-pass path through the network-size
    # fp32 masks of the reference API: identical masks for the same detections (bit-identical kernels, tests/test_kernels_gpu.py)
    two = mks()
    two.fused_masks = False
    assert one.fused_masks
    with torch.no_grad():
        r2 = [two.run(f, info) for f in fr]
    assert [r[0] for r in r1] == [r[0] for r in r2]
    same2 = sum(int(a == b) for x, y in zip(r1, r2) for a, b in zip(x[1], y[1]))
    assert same2 >= 0.9 * sum(len(r[1]) for r in r1), same2      # (the network outputs themselves carry fp64-atomic order noise between runs)
    del mm
    # ---- SOT
    ms, _, _ = build("unicorn_track_tiny", "f16x2")
    xywh = [float(box[0]), float(box[1]), float(box[2] - box[0]), float(box[3] - box[1])]
    a, b = UnicornSOTTrack(ms, input_size=(H, W)), UnicornSOTTrack(ms, input_size=(H, W))
    a.initialize(frames[0].cuda(), {"init_bbox": xywh})
    b.initialize(frames[0].cuda(), {"init_bbox": xywh})
    g1 = [a.track(f)["target_bbox"] for f in fr]
    g2 = [r["target_bbox"] for r in b.track_stream(fr)]
    assert all(abs(x - y) <= 1 for p_, q_ in zip(g1, g2) for x, y in zip(p_, q_)) and a.frame_id == b.frame_id == 6
    c = UnicornSOTTrack(ms, input_size=(H, W))
    c.initialize(frames[0].cuda(), {"init_bbox": xywh})
    g3 = [r["target_bbox"] for r in c.track_stream(fr, batch=4)]        # passes of 4 + 2 frames
    assert len(g3) == 6 and all(abs(x - y) <= 1 for p_, q_ in zip(g1, g3) for x, y in zip(p_, q_)) and c.frame_id == 6


def test_saturation_check_mode_counts_planted_outliers():
    """The f16x2 operand format saturates at +-65504 (csrc/common.h h2_split): no inf / NaN ever reaches an MFMA, and a context in check
    mode (uni_ctx_set_check) counts saturated operands.  (a) synthetic weights: operand buffers are scanned and nothing saturates, the
    outputs equal the unchecked run; (b) one ConvNeXt block with pwconv1 scaled by 1e6 (GELU hidden activations far beyond the f16
    range): the counter is raised, the outputs stay finite."""
    name, H, W = "unicorn_track_tiny", 320, 320
    cfg = uo.CONFIGS[name]
    P = synth.synth_state_dict(cfg)
    frames, _ = synth.synth_clip(H, W, 2, seed=1)
    from unicorn_amd.models import Unicorn
    m = Unicorn(name).cuda(0)
    assert m.precision == "f16x2"                                  # the parity mode is the default
    m.load_state_dict(P)
    with torch.no_grad():
        ref_fpn, _ = m(imgs=frames[1].cuda(), mode="backbone")
        m.check_saturation(True)
        fpn, _ = m(imgs=frames[1].cuda(), mode="backbone")
    st = m.saturation_stats()
    assert st["saturated"] == 0 and st["scanned"] > 1e6 and st["buffers"] > 50, st
    for a, b in zip(fpn, ref_fpn):
        assert (a - b).abs().max() < 1e-4 * b.abs().max()          # check mode only changes the launch plan (two-launch MLP)
    P2 = {k: v.clone() for k, v in P.items()}
    P2["backbone.backbone.stages.1.0.pwconv1.weight"] *= 1e6
    m2 = Unicorn(name).cuda(0)
    m2.load_state_dict(P2)
    m2.check_saturation(True)
    with torch.no_grad():
        fpn2, _ = m2(imgs=frames[1].cuda(), mode="backbone")
    st2 = m2.saturation_stats()
    assert st2["saturated"] > 0, st2
    assert all(torch.isfinite(t).all() for t in fpn2)
    m2.check_saturation(False)


def test_validate_checkpoint_tool_on_the_gpu(tmp_path, capsys):
    """tools/validate_checkpoint.py end to end: a `{"model": state_dict}` file -> loader rule of the reference ->
    HIP path in saturation check mode vs the CPU oracle -> parity block + saturation statistics + exit status.  (a) the synthetic
    weights of a mask model pass; (b) the same file with one pwconv1 scaled by 1e6 saturates f16x2 operands and must FAIL.
    The CPU half (oracle held to the real reference on a real-reference checkpoint) is tests/test_validate_checkpoint_cpu.py."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("validate_checkpoint", os.path.join(ROOT, "tools", "validate_checkpoint.py"))
    vc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(vc)
    name = "unicorn_track_tiny_mask"
    P = synth.synth_state_dict(uo.CONFIGS[name])
    good, bad = str(tmp_path / "good.pth"), str(tmp_path / "bad.pth")
    torch.save({"model": P, "start_epoch": 3}, good)
    rc = vc.main(["--ckpt", good, "--exp", name, "--size", "320x320", "--frames", "1", "--json", str(tmp_path / "good.json")])
    capsys.readouterr()
    rep = json.load(open(str(tmp_path / "good.json")))
    assert rc == 0 and rep["pass"] and rep["parity"]["pass"], rep
    assert rep["saturation"]["saturated"] == 0 and rep["saturation"]["scanned"] > 1e6, rep["saturation"]
    assert rep["parity"]["box_iou_min"] > 0.999 and rep["parity"]["embed_cos_min"] > 1 - 1e-4 and rep["parity"]["mask_iou_min"] > 0.999, rep["parity"]
    METRICS["validate_checkpoint_tiny_mask"] = rep["parity"]
    _dump()
    P2 = {k: v.clone() for k, v in P.items()}
    P2["backbone.backbone.stages.1.0.pwconv1.weight"] *= 1e6
    torch.save({"model": P2}, bad)
    rc = vc.main(["--ckpt", bad, "--exp", name, "--size", "320x320", "--frames", "1", "--json", str(tmp_path / "bad.json")])
    capsys.readouterr()
    rep = json.load(open(str(tmp_path / "bad.json")))
    assert rc == 1 and not rep["pass"] and rep["saturation"]["saturated"] > 0, rep.get("saturation")


def test_bench_two_ranks_shared_gpu_mix_plumbing():
    """The N > 1 path of bench.py on the ONE GPU of the test box: `UNI_BENCH_SHARE_GPU=1 bench.py --gpus 2 --task mix` self-launches
    two ranks (rank 0 the evaluate_omni MOT loop, rank 1 the SOT step, both on GPU 0), the RCCL probe must FAIL cleanly ("Duplicate GPU": two ranks on
    one device is not a configuration RCCL supports) and say why, the gathers fall back to gloo, result rows and RLE strings round-trip, no rank
    error.  This is rank assignment + fallback + gather plumbing, re-run by every driver GPUTEST -- NOT a scaling number (none is claimed)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(UNI_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--task", "mix", "--model", "unicorn_track_tiny", "--height", "320",
                          "--width", "512", "--batch", "2", "--steps", "4", "--warmup", "1", "--gather-every", "2", "--no-cpu-baseline", "--no-extras",
                          "--no-single-frame"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["value"] > 0 and not r.get("rank_errors"), r.get("rank_errors")
    tasks = [t["task"] for t in r["config"]["rank_tasks"]]
    assert tasks == ["mot", "sot"], tasks
    g = r["config"]["gather"]
    assert g["dist"]["data"] == "gloo" and g["dist"]["rccl_probe"]["ok_all_ranks"] is False, g["dist"]
    assert "why" in g["dist"]["rccl_probe"] and g["dist"]["rccl_probe"]["why"], g["dist"]["rccl_probe"]      # the one-GPU guard names the reason
    assert g["rows"] > 0 and g["lost_rows"] == 0 and g["rle_round_trip_ok"] and g["rle_undecodable"] == 0, g
    METRICS["bench_two_ranks_shared_gpu"] = {"tasks": tasks, "gather": {k: g[k] for k in ("calls", "rows", "lost_rows", "rle_strings")},
                                              "rccl_probe": g["dist"]["rccl_probe"]}
    _dump()


@pytest.mark.parametrize("exp", ["unicorn_track_tiny", "unicorn_track_large"])
def test_sot_driver_vs_reference_driver_class_golden(exp, golden_dir):
    """Drop-in evidence at the DRIVER level: tests/golden/driver_sot_*.npz hold what the reference's OWN `UnicornSOTTrack`
    (external/lib/test/tracker/unicorn_sot.py, imported unmodified, driven like lib/test/evaluation/tracker.py:138-198: initialize, then track per
    frame) produced with the reference's own model on the CPU (tests/golden/make_golden_drivers.py: raw uint8 RGB frames at the network size, planted
    detector-like scores, ~21000 candidates through postprocess = the largest NMS the path can see).  `unicorn_amd.tracker.UnicornSOTTrack` + the HIP
    model on the same frames must report the same detections (IoU > 0.999, scores within 1e-4) and the same integer `target_bbox` (+-1 px: the
    reference truncates floats)."""
    from planted import confident_head
    from unicorn_amd.models import Unicorn
    from unicorn_amd.tracker import UnicornSOTTrack
    H, W = 800, 1280
    g = np.load(os.path.join(golden_dir, "driver_sot_%s_%dx%d.npz" % (exp, H, W)))
    cfg = uo.CONFIGS[exp]
    P = confident_head(synth.synth_state_dict(cfg))
    m = Unicorn(exp, precision="f16x2").cuda()
    assert not m.load_state_dict(P, strict=False)[0]
    m.eval()
    n, seed = int(g["nframes"][0]), int(g["seed"][0])
    rng = np.random.default_rng(seed)                                  # tests/golden/make_golden_drivers.py:driver_clip
    base = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    frames = []
    for t in range(n + 1):
        f = np.roll(base, (3 * t, 5 * t), (0, 1)).astype(np.int16) + rng.integers(0, 8, (H, W, 3), dtype=np.int16)
        frames.append(np.clip(f, 0, 255).astype(np.uint8))
    trk = UnicornSOTTrack(m, input_size=(H, W))
    trk.initialize(frames[0], {"init_bbox": [float(v) for v in g["init_bbox"]]})
    cx = lambda t: torch.stack([(t[:, 0] + t[:, 2]) / 2, (t[:, 1] + t[:, 3]) / 2, t[:, 2] - t[:, 0], t[:, 3] - t[:, 1]], 1)
    met = []
    for t in range(1, n + 1):
        ref = torch.from_numpy(g["det_%d" % t])
        cur, _ = trk._prep(frames[t])
        det = trk.get_det_results(cur)
        assert det is not None
        nd = int(det.shape[0])
        d = det[:ref.shape[0]].cpu().clone()
        d[:, 0:4:2] = d[:, 0:4:2].clamp(min=0, max=W)
        d[:, 1:4:2] = d[:, 1:4:2].clamp(min=0, max=H)
        iou = box_iou_pairs(cx(d[:, :4]), cx(ref[:, :4]))
        res = trk.track(frames[t])
        state_ref = g["target_bbox_%d" % t]
        met.append({"frame": t, "n_det": [nd, int(g["n_det_%d" % t][0])], "iou_min": float(iou.min()),
                    "score_maxabs": float((d[:, 4] * d[:, 5] - ref[:, 4] * ref[:, 5]).abs().max()),
                    "target_bbox": [[int(v) for v in res["target_bbox"]], [int(v) for v in state_ref]]})
        assert float(iou.min()) > 0.999 and met[-1]["score_maxabs"] < 1e-4, met[-1]
        assert abs(nd - int(g["n_det_%d" % t][0])) <= max(2, nd // 2000), met[-1]      # NMS survivors: borderline IoU pairs may flip among ~21000
        assert max(abs(int(a) - int(b)) for a, b in zip(res["target_bbox"], state_ref)) <= 1, met[-1]
    METRICS["driver_sot_%s" % exp] = met
    _dump()


def test_omni_loop_vs_reference_evaluate_omni_method_golden(golden_dir):
    """tests/golden/driver_omni_*.npz = what the reference's OWN `MOTEvaluator.evaluate_omni` METHOD (mot_evaluator.py:925-1105, executed unmodified on a stand-in
    dataloader with the reference model and the reference `QuasiDenseEmbedTracker()` defaults, tests/golden/make_golden_drivers.py) wrote to its MOT result file,
    plus the boxes / ids its tracker returned per frame.  `unicorn_amd.tracker.OmniMOTFrame` + native tracker + HIP model on the same frames: the same ~300
    detections per frame reach the tracker (boxes within 0.05 px at the original-image scale, scores within 1e-5), the same ids come back, and the filtered, rounded
    result rows (`:1064-1082`, `write_results` `:49-58`) are the file's."""
    from planted import confident_head
    from unicorn_amd.models import Unicorn
    from unicorn_amd.tracker import OmniMOTFrame, QuasiDenseEmbedTracker
    exp, H, W = "unicorn_track_large_mot_challenge", 800, 1280
    g = np.load(os.path.join(golden_dir, "driver_omni_%s_%dx%d.npz" % (exp, H, W)))
    cfg = uo.CONFIGS[exp]
    P = confident_head(synth.synth_state_dict(cfg), 0.0, 0.0, 2.0)
    m = Unicorn(exp, precision="f16x2").cuda()
    assert not m.load_state_dict(P, strict=False)[0]
    m.eval()
    n, seed = int(g["nframes"][0]), int(g["seed"][0])
    frames, _ = synth.synth_clip(H, W, n + 1, seed=seed)
    img_hw = tuple(int(v) for v in g["img_hw"])
    trk = QuasiDenseEmbedTracker()                                      # the reference's defaults (evaluate_omni :968, :979)
    seen, inner = [], trk.match

    def tap(bboxes, labels, feats, frame_id, *a, **k):
        r = inner(bboxes, labels, feats, frame_id, *a, **k)
        seen.append((torch.as_tensor(r[0]).clone(), torch.as_tensor(r[2]).clone()))
        return r
    trk.match = tap
    omni = OmniMOTFrame(m, trk, (H, W), num_classes=1, confthre=float(g["confthre"][0]), nmsthre=0.7, embed_score_thr=0.1)
    rows, met = [], []
    for t in range(1, n + 1):
        with torch.no_grad():
            out_b, out_ids = omni.run(frames[t].cuda(), img_hw)
        b_ref, i_ref = torch.from_numpy(g["match_bboxes_%d" % t]), torch.from_numpy(g["match_ids_%d" % t])
        b_h, i_h = seen[-1]
        assert abs(b_h.shape[0] - b_ref.shape[0]) <= 2, (t, b_h.shape, b_ref.shape)
        if b_h.shape[0] == b_ref.shape[0]:
            d = (b_h[:, :4] - b_ref[:, :4]).abs().max(1)[0]
            same_order = bool((d < 0.05).all())
            if not same_order:      # two near-equal scores may swap rows: compare as sets
                d = torch.stack([(b_ref[:, :4] - bh[:4]).abs().max(1)[0].min() for bh in b_h])
            met.append({"frame": t, "n": int(b_h.shape[0]), "box_maxabs": float(d.max()), "same_order": same_order,
                        "score_maxabs": float((b_h[:, 4] - b_ref[:, 4]).abs().max()) if same_order else None})
            assert float(d.max()) < 0.05, met[-1]
            if same_order:
                assert met[-1]["score_maxabs"] < 1e-5 and torch.equal(i_h.long(), i_ref.long()), met[-1]
        assert sorted(int(v) for v in i_ref[i_ref > -1]) == sorted(int(v) for v in torch.as_tensor(out_ids)), (t, out_ids, i_ref[i_ref > -1])
        ob = torch.as_tensor(out_b).numpy()
        for i in range(ob.shape[0]):                                     # evaluate_omni :1064-1082
            x1, y1, x2, y2, score = [float(v) for v in ob[i]]
            w, h = x2 - x1, y2 - y1
            if w * h > 10 and not (w / h > 1.6):
                rows.append([t, int(torch.as_tensor(out_ids)[i]) + 1, round(x1, 1), round(y1, 1), round(w, 1), round(h, 1), round(score, 2)])
    ref_rows = g["rows"]
    METRICS["driver_omni_%s" % exp] = {"frames": met, "result_rows": [len(rows), int(ref_rows.shape[0])]}
    _dump()
    assert len(rows) == ref_rows.shape[0], (rows, ref_rows)
    rr = np.array(rows, dtype=np.float64)
    assert np.array_equal(rr[:, :2], ref_rows[:, :2]) and np.abs(rr[:, 2:6] - ref_rows[:, 2:6]).max() <= 0.11 and np.abs(rr[:, 6] - ref_rows[:, 6]).max() <= 0.011, (rr, ref_rows)


def test_byte_loop_vs_reference_evaluate_method_golden(golden_dir):
    """tests/golden/driver_byte_*.npz = the MOT result file of the reference's OWN `MOTEvaluator.evaluate` method (mot_evaluator.py:100-240, what `tools/track.py`
    calls; reference model + reference `BYTETracker`, executed unmodified on a stand-in dataloader, tests/golden/make_golden_drivers.py:run_byte_evaluator) and the
    tracks its tracker returned per frame.  `unicorn_amd.tracker.ByteMOTFrame` + the native BYTETracker + the HIP model on the same frames: same track ids per frame,
    tlwh within 0.05 px, scores within 1e-5, and the same filtered, rounded result rows."""
    from types import SimpleNamespace
    from planted import confident_head
    from unicorn_amd.models import Unicorn
    from unicorn_amd.tracker import BYTETracker, ByteMOTFrame
    from unicorn_amd.tracker import byte_tracker as bt
    exp, H, W = "unicorn_track_large_mot_challenge", 800, 1280
    g = np.load(os.path.join(golden_dir, "driver_byte_%s_%dx%d.npz" % (exp, H, W)))
    cfg = uo.CONFIGS[exp]
    P = confident_head(synth.synth_state_dict(cfg), 0.0, 0.0, 2.0)
    m = Unicorn(exp, precision="f16x2").cuda()
    assert not m.load_state_dict(P, strict=False)[0]
    m.eval()
    n, seed = int(g["nframes"][0]), int(g["seed"][0])
    base, _ = synth.synth_clip(H, W, 2, seed=seed)                      # make_golden_drivers.py:byte_clip
    frames = [base[0]] + [torch.roll(base[1], shifts=max(t - 2, 0), dims=3).contiguous() for t in range(1, n + 1)]
    info = (int(g["img_hw"][0]), int(g["img_hw"][1]), 1, 1, "SYN-02/img1/000001.jpg")
    args = SimpleNamespace(track_thresh=float(g["track_thresh"][0]), track_buffer=30, match_thresh=0.9, mot20=False)
    bt.clean_id()
    trk = BYTETracker(args)
    seen, inner = [], trk.update

    def tap(*a, **k):
        r = inner(*a, **k)
        seen.append(np.array([[*t_.tlwh, t_.track_id, t_.score] for t_ in r], dtype=np.float64).reshape(-1, 6))
        return r
    trk.update = tap
    byte = ByteMOTFrame(m, trk, (H, W), num_classes=1, confthre=float(g["confthre"][0]), nmsthre=0.7, min_box_area=10)
    rows, met = [], []
    for t in range(1, n + 1):
        with torch.no_grad():
            res = byte.run(frames[t].cuda(), info)
        ref = g["tracks_%d" % t]
        got = seen[-1]
        assert got.shape == ref.shape, (t, got.shape, ref.shape)
        o_g, o_r = np.argsort(got[:, 4], kind="stable"), np.argsort(ref[:, 4], kind="stable")
        got, ref = got[o_g], ref[o_r]
        met.append({"frame": t, "tracks": int(ref.shape[0]), "tlwh_maxabs": float(np.abs(got[:, :4] - ref[:, :4]).max()) if ref.size else 0.0,
                    "score_maxabs": float(np.abs(got[:, 5] - ref[:, 5]).max()) if ref.size else 0.0})
        assert np.array_equal(got[:, 4], ref[:, 4]), (t, got[:, 4], ref[:, 4])
        assert met[-1]["tlwh_maxabs"] < 0.05 and met[-1]["score_maxabs"] < 1e-5, met[-1]
        if res is not None:
            for tlwh, tid, s_ in zip(*res):                             # write_results (:49-58)
                rows.append([t, tid, round(float(tlwh[0]), 1), round(float(tlwh[1]), 1), round(float(tlwh[2]), 1), round(float(tlwh[3]), 1), round(float(s_), 2)])
    ref_rows = g["rows"]
    METRICS["driver_byte_%s" % exp] = {"frames": met, "result_rows": [len(rows), int(ref_rows.shape[0])]}
    _dump()
    rr = np.array(rows, dtype=np.float64).reshape(-1, 7)
    assert rr.shape == ref_rows.shape, (rr.shape, ref_rows.shape)
    key = lambda a: np.lexsort((a[:, 1], a[:, 0]))
    rr, ref_rows = rr[key(rr)], ref_rows[key(ref_rows)]
    assert np.array_equal(rr[:, :2], ref_rows[:, :2]) and np.abs(rr[:, 2:6] - ref_rows[:, 2:6]).max() <= 0.11 and np.abs(rr[:, 6] - ref_rows[:, 6]).max() <= 0.011, (rr, ref_rows)


def test_vos_driver_vs_reference_driver_class_golden(golden_dir):
    """tests/golden/driver_vos_*.npz = the id maps and per-object integer box states the reference's OWN `UnicornVOSTrack` class
    (external/lib/test/tracker/unicorn_vos.py, unmodified, reference model, CPU; tests/golden/make_golden_drivers.py:run_vos_driver) produced for K = 3 objects over two
    frames at 800 x 1280.  `unicorn_amd.tracker.UnicornVOSTrack` (one correlation + one object-batched head call per group, device-side soft aggregation) + the HIP
    model on the same uint8 frames: the same box states (+-1 px) and the same id map up to pixels whose object probabilities tie (the synthetic mask heads put large
    areas at p ~ 0.5; agreement is logged -- 0.999999 measured, one pixel of 1 024 000 -- and held above 0.9999)."""
    from planted import confident_head
    from unicorn_amd.models import Unicorn
    from unicorn_amd.tracker import UnicornVOSTrack
    exp, H, W = "unicorn_track_tiny_mask", 800, 1280
    g = np.load(os.path.join(golden_dir, "driver_vos_%s_%dx%d.npz" % (exp, H, W)))
    cfg = uo.CONFIGS[exp]
    P = confident_head(synth.synth_state_dict(cfg), -4.2, -4.2)
    m = Unicorn(exp, precision="f16x2").cuda()
    assert not m.load_state_dict(P, strict=False)[0]
    m.eval()
    n, seed = int(g["nframes"][0]), int(g["seed"][0])
    rng = np.random.default_rng(seed)                                  # tests/golden/make_golden_drivers.py:driver_clip
    base = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    frames = []
    for t in range(n + 1):
        f = np.roll(base, (3 * t, 5 * t), (0, 1)).astype(np.int16) + rng.integers(0, 8, (H, W, 3), dtype=np.int16)
        frames.append(np.clip(f, 0, 255).astype(np.uint8))
    ids = ["1", "2", "3"]
    boxes = {k: [float(v) for v in g["boxes"][i]] for i, k in enumerate(ids)}
    trk = UnicornVOSTrack(m, input_size=(H, W), d_rate=cfg.d_rate)
    trk.initialize(frames[0], {"init_object_ids": list(ids), "sequence_object_ids": list(ids), "init_bbox": boxes})
    met = []
    for t in range(1, n + 1):
        seg = trk.track(frames[t], {})["segmentation"]
        ref = g["seg_%d" % t]
        assert seg.shape == ref.shape and seg.dtype == np.uint8
        states = np.array([trk.state_pre_dict[k] for k in ids], dtype=np.float64)
        met.append({"frame": t, "agreement": float((seg == ref).mean()), "states_maxabs": float(np.abs(states - g["states_%d" % t]).max())})
    METRICS["driver_vos_%s" % exp] = met
    _dump()
    for r in met:
        assert r["states_maxabs"] <= 1.0 and r["agreement"] > 0.9999, met


def test_c_host_without_torch_matches_the_python_path(tmp_path):
    """The drop-in boundary is a C-ABI library, not a torch extension: `tools/build/capi_host_demo` (tools/capi_host_demo.cpp: include/unicorn_hip.h + the HIP runtime,
    nothing else -- no Python, no torch) loads a flat weights file (uni_weights_file_cfg / uni_ctx_load_file, written by unicorn_amd.utils.checkpoint.export_flat: the
    artefact in the role of tools/export_torchscript.py:51-71), runs the SOT step of unicorn_sot.py:39-55,78-108 on two hash-pattern frames and writes the raw head rows.
    The Python path on the same frames (weights through the same flat file, `Unicorn.load_flat_file`, and once more through load_state_dict) must give the same rows."""
    import subprocess
    from unicorn_amd.models import Unicorn
    from unicorn_amd.ops import corr_softmax_pv, label_map_s8, prior_pyramid
    from unicorn_amd.utils.checkpoint import export_flat
    demo = os.path.join(ROOT, "tools", "build", "capi_host_demo")
    if not os.path.exists(demo):
        pytest.fail("tools/build/capi_host_demo is missing: run __graft_entry__.build() (csrc/build.sh builds it)")
    exp, H, W = "unicorn_track_tiny", 320, 512
    cfg = uo.CONFIGS[exp]
    P = synth.synth_state_dict(cfg)
    wfile, ofile = str(tmp_path / "tiny.uniw"), str(tmp_path / "rows.bin")
    export_flat(P, exp, wfile)
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTHON")}
    cp = subprocess.run([demo, wfile, str(H), str(W), ofile], capture_output=True, text=True, timeout=300, env=env)
    assert cp.returncode == 0, (cp.stdout[-500:], cp.stderr[-1500:])
    info = json.loads(cp.stdout.strip().splitlines()[-1])
    rows_c = torch.from_numpy(np.fromfile(ofile, dtype=np.float32).reshape(-1, 6))
    # the same frames in numpy: ((i + 977 t) * 2654435761 mod 2^32) >> 24
    idx = np.arange(3 * H * W, dtype=np.uint64)
    frame = lambda t: torch.from_numpy((((idx + np.uint64(977 * t)) * np.uint64(2654435761)) % np.uint64(1 << 32) >> np.uint64(24)).astype(np.float32)).view(1, 3, H, W)
    box = torch.tensor([W * 0.25, H * 0.25, W * 0.5, H * 0.5])

    def python_rows(m):
        with torch.no_grad():
            _, d_pre = m(imgs=frame(0).cuda(), mode="backbone")
            lbs = label_map_s8(box, H, W, "cuda")
            fpn, d_cur = m(imgs=frame(1).cuda(), mode="backbone")
            f_pre, f_cur = m(seq_dict0=d_pre, seq_dict1=d_cur, mode="interaction")
            e_pre, e_cur = m(feat=f_pre, mode="upsample"), m(feat=f_cur, mode="upsample")
            pred = corr_softmax_pv(e_pre.flatten(-2).squeeze(0), e_cur.flatten(-2).squeeze(0), lbs)
            coarse = pred.view(1, -1, d_cur["h"] * 2, d_cur["w"] * 2)
            return m.head(fpn, prior_pyramid(coarse), mode="sot")[0].cpu()
    m1 = Unicorn(exp, precision="f16x2").cuda()
    assert m1.load_flat_file(wfile) == info["tensors_loaded"]
    rows_f = python_rows(m1)
    m2 = Unicorn(exp, precision="f16x2").cuda()
    m2.load_state_dict(P)
    rows_s = python_rows(m2)
    assert rows_c.shape == rows_f.shape == rows_s.shape and info["anchors"] == rows_c.shape[0]
    scale = float(rows_s.abs().max())
    err_c, err_f = float((rows_c - rows_s).abs().max()) / scale, float((rows_f - rows_s).abs().max()) / scale
    METRICS["c_host_without_torch"] = {"rows": int(rows_c.shape[0]), "max_rel_diff_c_host_vs_python": err_c, "max_rel_diff_flat_file_vs_state_dict": err_f,
                                       "best_anchor": info["best_anchor"], "demo_stdout": info}
    _dump()
    # same kernels, same weights: only the fp64-atomic order of the GroupNorm sums may differ between runs
    assert err_c < 2e-5 and err_f < 2e-5, (err_c, err_f)
    assert int(torch.argmax(rows_s[:, 4] * rows_s[:, 5])) == info["best_anchor"]


def test_round6_entry_points_edge_cases(tmp_path):
    """Edge cases of the round-6 entry points: ByteMOTFrame on a time batch (B = 2 consecutive frames per call = two calls) and on frames WITHOUT detections (the
    tracker is not stepped, mot_evaluator.py:211); DemoPredictor without detections (`postprocess` -> [None]); the flat weights file loader on a file for ANOTHER
    network configuration and on a truncated file (error strings, no crash); `Unicorn.load_flat_file` after weights were loaded."""
    import ctypes as C
    from types import SimpleNamespace
    from planted import confident_head
    from unicorn_amd import _lib as L
    from unicorn_amd.models import Unicorn
    from unicorn_amd.tracker import BYTETracker, ByteMOTFrame, DemoPredictor
    from unicorn_amd.tracker import byte_tracker as bt
    from unicorn_amd.utils.checkpoint import export_flat
    exp, H, W = "unicorn_track_tiny", 320, 320
    cfg = uo.CONFIGS[exp]
    P = confident_head(synth.synth_state_dict(cfg))
    m = Unicorn(exp, precision="f16x2").cuda()
    m.load_state_dict(P)
    frames, _ = synth.synth_clip(H, W, 3, seed=9)
    with torch.no_grad():
        o, _ = m(frames[1].cuda())
    sc = (o[0, :, 4] * o[0, :, 5:].max(1)[0]).sort(descending=True)[0]
    args = SimpleNamespace(track_thresh=float((sc[39] + sc[40]) / 2), track_buffer=30, match_thresh=0.9, mot20=False)
    info = (480, 480, 1, 1, "x")
    conf = float((sc[99] + sc[100]) / 2)
    bt.clean_id()
    seq = ByteMOTFrame(m, BYTETracker(args), (H, W), num_classes=cfg.num_classes, confthre=conf, nmsthre=0.7, min_box_area=1)
    with torch.no_grad():
        r_seq = [seq.run(frames[t].cuda(), info) for t in (1, 2)]
    bt.clean_id()
    bat = ByteMOTFrame(m, BYTETracker(args), (H, W), num_classes=cfg.num_classes, confthre=conf, nmsthre=0.7, min_box_area=1)
    with torch.no_grad():
        r_bat = bat.run_batch(torch.cat([frames[1], frames[2]], 0).cuda(), info)
    assert len(r_bat) == 2
    for a, b in zip(r_seq, r_bat):
        assert (a is None) == (b is None)
        if a is not None:
            assert list(a[1]) == list(b[1]) and all(float(np.abs(x - y).max()) < 1e-2 for x, y in zip(a[0], b[0]))      # B = 1 / B = 2 take different tile shapes
    none = ByteMOTFrame(m, BYTETracker(args), (H, W), num_classes=cfg.num_classes, confthre=2.0, nmsthre=0.7)
    with torch.no_grad():
        assert none.run(frames[1].cuda(), info) is None and none.tracker.frame_id == 0                                       # tracker not stepped
    dets, _ = DemoPredictor(m, cfg.num_classes, 2.0, 0.45, (H, W)).inference(np.zeros((200, 300, 3), dtype=np.uint8))
    assert dets == [None]
    # ---- flat weights file: wrong configuration, truncated file, double load
    good, other = str(tmp_path / "tiny.uniw"), str(tmp_path / "mask.uniw")
    export_flat(P, exp, good)
    cm = uo.CONFIGS["unicorn_track_tiny_mask"]
    export_flat(synth.synth_state_dict(cm), "unicorn_track_tiny_mask", other)
    m2 = Unicorn(exp, precision="f16x2").cuda()
    with pytest.raises(L.UnicornHipError, match="another network configuration"):
        m2.load_flat_file(other)
    cut = str(tmp_path / "cut.uniw")
    open(cut, "wb").write(open(good, "rb").read()[:200000])
    m3 = Unicorn(exp, precision="f16x2").cuda()
    with pytest.raises(L.UnicornHipError, match="truncated"):
        m3.load_flat_file(cut)
    with pytest.raises(L.UnicornHipError, match="already loaded"):
        m.load_flat_file(good)
    m4 = Unicorn(exp, precision="fp32").cuda()                         # the file records f16x2; the context's own precision wins
    assert m4.load_flat_file(good) > 500
    with torch.no_grad():
        o4, _ = m4(frames[1].cuda())
    assert float((o4 - o).abs().max()) / float(o.abs().max()) < 1e-4
