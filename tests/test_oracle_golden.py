"""Pin the oracle (oracle/unicorn_oracle.py) against golden vectors produced by the REAL reference
(tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

import synth
import unicorn_oracle as uo

MAX_FULL = 1 << 15


def sample(t, max_full=MAX_FULL):
    a = t.detach().float().contiguous().numpy().reshape(-1)
    if a.size <= max_full:
        return a
    return a[np.linspace(0, a.size - 1, max_full).astype(np.int64)]


# Oracle and reference are both fp32 torch-CPU programs: the backbone / FPN maps come out BIT-IDENTICAL, everything behind the interaction
# within 4e-6 of the largest element (measured over all golden cases, round 6: worst `coarse` 3.5e-6, `dyn_params` 2.4e-6).  The bar is
# 1e-5 x scale -- tight enough to catch a wrong eps or a dropped bias that the 2e-4 of rounds 1-5 could hide behind a large-magnitude map.
TOL = 1e-5


def check(g, name, t, tol=TOL):
    assert tuple(g[name + "__shape"]) == tuple(t.shape), name
    ref = g[name]
    got = sample(t, int(g["__max_full"][0]) if "__max_full" in g.files else MAX_FULL)
    scale = max(1.0, float(np.abs(ref).max()))
    err = np.abs(got - ref).max()
    assert err <= tol * scale, "%s: max err %g (scale %g)" % (name, err, scale)


@pytest.mark.parametrize("exp", ["unicorn_track_tiny", "unicorn_track_tiny_mask", "unicorn_track_large",
                                 "unicorn_track_large_mask", "unicorn_track_large_mot_challenge"])
def test_param_spec_matches_reference_layout(exp, golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "state_spec_%s.json" % exp)))
    spec = synth.param_spec(uo.CONFIGS[exp])
    assert set(spec) == set(ref)
    for k, v in spec.items():
        assert list(v) == ref[k], k


def test_msda_known_answer(golden_dir):
    """unicorn/models/ops/test.py:24-50 (shapes, seed 3) + a bigger case with out-of-range samples."""
    g = np.load(os.path.join(golden_dir, "msda_known_answer.npz"))
    for sfx in ("", "2"):
        shapes = [tuple(int(v) for v in r) for r in g["shapes" + sfx]]
        out = uo.msda_core(torch.from_numpy(g["value" + sfx]), shapes, torch.from_numpy(g["loc" + sfx]),
                           torch.from_numpy(g["attn" + sfx]))
        ref = torch.from_numpy(g["out" + sfx])
        assert torch.allclose(out, ref, rtol=1e-4, atol=1e-6), (out - ref).abs().max()


@pytest.mark.parametrize("exp,H,W", [("unicorn_track_tiny", 320, 320), ("unicorn_track_tiny_mask", 320, 320), ("unicorn_track_large", 320, 320),
                                     ("unicorn_track_large_mask", 320, 320), ("unicorn_track_large_mot_challenge", 320, 320),
                                     ("unicorn_track_tiny_mask", 320, 512)])      # non-square: the 800 x 1280 aspect (H / W = 0.625)
def test_sot_step_matches_reference(exp, H, W, golden_dir):
    """BASELINE.json configs[0]: tiny, 2-frame 320x320 synthetic clip, CPU -- and the same clip through the REAL reference's
    headline model (`unicorn_track_large`, depths [3,3,27,3] / dims [192..1536], convnext.py:198-211), its mask variant and the
    num_classes = 1 MOT-challenge head (exps/default/unicorn_track_large_mot_challenge.py:18)."""
    torch.set_num_threads(8)
    g = np.load(os.path.join(golden_dir, "%s_%dx%d.npz" % (exp, H, W)))
    cfg = uo.CONFIGS[exp]
    P = synth.synth_state_dict(cfg)
    frames, box = synth.synth_clip(H, W, 2, seed=1)
    with torch.no_grad():
        st = uo.sot_init(P, cfg, frames[0], box)
        check(g, "lbs_pre", st["lbs_pre"])
        r = uo.sot_step(P, cfg, st, frames[1])
        for i in range(3):
            check(g, "fpn%d" % i, r["fpn"][i])
        check(g, "seq_feat", r["seq"]["feat"])
        check(g, "seq_pos", r["seq"]["pos"])
        check(g, "feat_pre", r["feat_pre"])
        check(g, "feat_cur", r["feat_cur"])
        check(g, "embed_pre", r["embed_pre"])
        check(g, "embed_cur", r["embed_cur"])
        check(g, "coarse", r["coarse"])
        pri = uo.prior_pyramid(r["coarse"])
        check(g, "prior16", pri[1])
        check(g, "prior32", pri[2])
        if cfg.mask:
            names = ["head_out", "locations", "dyn_params", "fpn_levels", "mask_feats", "up_masks"]
            for n, t in zip(names, r["head"]):
                check(g, n, t)
            head = tuple(t.clone() for t in r["head"])
            det, masks = uo.postprocess_inst(cfg, head, 1, 0.001, 0.65)
            assert det.shape[0] == int(g["n_det_sot"][0])
            check(g, "det_sot", det[:8])
            bits = np.packbits((masks[:8] > 0.5).numpy().astype(np.uint8).reshape(-1))
            ref_bits = g["mask_sot_bits"]
            diff = np.unpackbits(bits ^ ref_bits).sum()
            assert diff <= 1e-5 * ref_bits.size * 8, diff
            check(g, "mask_sot", masks[:8][:, :, ::8, ::8], tol=5e-5)      # sigmoid of the dynamic-MLP logits (|logit| up to ~30): 1.4e-5 measured
        else:
            check(g, "head_out", r["head"])
            det = uo.postprocess(r["head"].clone(), 1, 0.001, 0.65)[0]
            assert det.shape[0] == int(g["n_det_sot"][0])
            check(g, "det_sot", det[:64])
        whole, _, _ = uo.mot_whole(P, cfg, frames[1])
        who = whole[0] if cfg.mask else whole
        check(g, "whole_out", who)
        det = uo.postprocess(who.clone(), cfg.num_classes, 0.0005, 0.65)[0]
        assert (0 if det is None else det.shape[0]) == int(g["n_det_mot"][0])
        if "det_mot" in g.files:
            check(g, "det_mot", det[:64])
            emb = uo.sample_instance_embeddings(r["embed_cur"], det[:16, :4])
            check(g, "inst_embed", emb)


@pytest.mark.parametrize("exp", ["unicorn_track_large", "unicorn_track_large_mask"])
def test_headline_800x1280_matches_reference(exp, golden_dir):
    """The HEADLINE configuration (exp/unicorn_track.py:104: test_size (800, 1280)) through the REAL reference on the CPU
    (tests/golden/make_golden.py:run_headline): 50 x 80 token grid, pos-embed resized UP (40 -> 50 / 80), the 16000 x 16000 correlation, the
    C = 1536 25 x 40 maps -- the oracle that the GPU tests and bench.py's in-run parity lean on at this size is held to the reference AT
    this size, not only at 320 x 320.  Raw head rows of the 500 best anchors (SOT head and mode="whole") are compared row by row."""
    torch.set_num_threads(8)
    H, W = 800, 1280
    g = np.load(os.path.join(golden_dir, "%s_%dx%d.npz" % (exp, H, W)))
    cfg = uo.CONFIGS[exp]
    P = synth.synth_state_dict(cfg)
    frames, box = synth.synth_clip(H, W, 2, seed=1)
    with torch.no_grad():
        st = uo.sot_init(P, cfg, frames[0], box)
        check(g, "lbs_pre", st["lbs_pre"])
        r = uo.sot_step(P, cfg, st, frames[1])
        for i in range(3):
            check(g, "fpn%d" % i, r["fpn"][i])
        for k, t in (("seq_feat", r["seq"]["feat"]), ("seq_pos", r["seq"]["pos"]), ("feat_pre", r["feat_pre"]), ("feat_cur", r["feat_cur"]),
                     ("embed_pre", r["embed_pre"]), ("embed_cur", r["embed_cur"]), ("coarse", r["coarse"])):
            check(g, k, t)
        pri = uo.prior_pyramid(r["coarse"])
        check(g, "prior16", pri[1])
        check(g, "prior32", pri[2])
        head = r["head"][0] if cfg.mask else r["head"]
        check(g, "head_out", head)
        top = torch.from_numpy(g["head_top_idx"])
        ref = torch.from_numpy(g["head_top_rows"])
        assert torch.equal(torch.argsort(head[0, :, 4] * head[0, :, 5], descending=True)[:50], top[:50])      # same ranking at the top
        assert (head[0, top] - ref).abs().max() <= TOL * float(ref.abs().max())
        if cfg.mask:
            for n, t in zip(["locations", "dyn_params", "fpn_levels", "mask_feats", "up_masks"], r["head"][1:]):
                check(g, n, t)
            assert (r["head"][2][0, top[:64]] - torch.from_numpy(g["dyn_top_rows"])).abs().max() <= TOL * float(np.abs(g["dyn_top_rows"]).max())
            det, masks = uo.postprocess_inst(cfg, tuple(t.clone() for t in r["head"]), 1, 0.001, 0.65)
            assert det.shape[0] == int(g["n_det_sot"][0])
            k = int(g["det_sot__shape"][0])
            check(g, "det_sot", det[:k])
            bits = np.packbits((masks[:k] > 0.5).numpy().astype(np.uint8).reshape(-1))
            assert np.unpackbits(bits ^ g["mask_sot_bits"]).sum() <= 1e-5 * g["mask_sot_bits"].size * 8
        else:
            det = uo.postprocess(head.clone(), 1, 0.001, 0.65)[0]
            assert det.shape[0] == int(g["n_det_sot"][0])
            check(g, "det_sot", det[:64])
        whole, _, _ = uo.mot_whole(P, cfg, frames[1])
        who = whole[0] if cfg.mask else whole
        check(g, "whole_out", who)
        wtop, wref = torch.from_numpy(g["whole_top_idx"]), torch.from_numpy(g["whole_top_rows"])
        assert (who[0, wtop] - wref).abs().max() <= TOL * float(wref.abs().max())


@pytest.mark.parametrize("exp", ["unicorn_track_tiny_mask", "unicorn_track_large_mask"])
def test_vos_step_matches_reference(exp, golden_dir):
    """BASELINE.json configs[3] in small: the VOS driver's get_det_results (unicorn_vos.py:157-200) for K = 3 objects, produced with the REAL
    reference model + its postprocess_inst (tests/golden/make_golden.py:run_vos): best instance and its CondInst mask per object."""
    torch.set_num_threads(8)
    H = W = 320
    g = np.load(os.path.join(golden_dir, "%s_vos_%dx%d.npz" % (exp, H, W)))
    cfg = uo.CONFIGS[exp]
    P = synth.synth_state_dict(cfg)
    frames, box = synth.synth_clip(H, W, 2, seed=1)
    boxes = {"1": box, "2": torch.tensor([W * 0.55, H * 0.1, W * 0.9, H * 0.45]), "3": torch.tensor([W * 0.1, H * 0.55, W * 0.4, H * 0.95])}
    with torch.no_grad():
        st = uo.vos_init(P, cfg, frames[0], boxes)
        res = uo.vos_step(P, cfg, st, frames[1])
    for k in boxes:
        det, mask = res[k]
        assert (det is None) == (int(g["n_det_%s" % k][0]) == 0), k
        if det is None:
            continue
        ref = torch.from_numpy(g["det_%s" % k])
        assert (det[:4] - ref[:4]).abs().max() < 2e-2 and (det[4:6] - ref[4:6]).abs().max() < 1e-5 * max(1.0, float(ref[4:6].abs().max())) + 1e-6, (k, det, ref)
        bits = np.packbits((mask > 0.5).numpy().astype(np.uint8).reshape(-1))
        diff = np.unpackbits(bits ^ g["mask_bits_%s" % k]).sum()
        assert diff <= 1e-4 * mask.numel(), (k, diff)
        check(g, "mask_%s" % k, mask[::4, ::4], tol=5e-5)


def test_letterbox_oracle_known_answers():
    """oracle/letterbox_oracle.py restates cv2.resize(INTER_LINEAR, uint8) (third-party, absent offline): properties every
    correct implementation has -- identity, constants, the 3:1 / 1:3 pattern of an exact 2x upsample, the rounded 2x2 mean of
    an exact 2x downsample -- plus the letterbox geometry of PreprocessorX.process (unicorn_sot.py:111-123)."""
    import letterbox_oracle as lo
    g = np.random.default_rng(0)
    img = g.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(lo.cv2_resize_linear_u8(img, (53, 37)), img)
    assert (lo.cv2_resize_linear_u8(np.full((20, 30, 3), 77, np.uint8), (91, 47)) == 77).all()
    ramp = np.arange(0, 80, 8, dtype=np.uint8).reshape(1, 10, 1).repeat(3, 2)
    assert lo.cv2_resize_linear_u8(ramp, (20, 1))[0, :, 0].tolist() == [0, 2, 6, 10, 14, 18, 22, 26, 30, 34, 38, 42, 46, 50, 54, 58,
                                                                       62, 66, 70, 72]
    blk = img[:36, :52].astype(np.int64).reshape(18, 2, 26, 2, 3).sum((1, 3))
    assert np.array_equal(lo.cv2_resize_linear_u8(img[:36, :52], (26, 18)), ((blk + 2) >> 2).astype(np.uint8))
    out, r = lo.letterbox(g.integers(0, 256, (1080, 1920, 3), dtype=np.uint8), (800, 1280), True)
    assert out.shape == (3, 800, 1280) and r == 800 / 1200 or r == min(800 / 1080, 1280 / 1920)
    assert (out[:, 720:, :] == 114).all() and out.dtype == np.float32


@pytest.mark.parametrize("A,nc,agn", [(2100, 1, False), (5000, 8, False), (5000, 8, True), (333, 3, False)])
def test_postprocess_oracle_matches_reference_golden(A, nc, agn):
    """oracle.postprocess against the REAL reference function (unicorn/utils/boxes.py:33-77; tests/golden/make_golden_post.py) on
    planted head outputs: same rows, same order (the torchvision NMS it calls is third-party, restated)."""
    from planted import planted_pred
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "postprocess_planted.npz"))
    det = uo.postprocess(planted_pred(A, nc, seed=A + nc), nc, 0.2, 0.45, class_agnostic=agn)[0]
    assert np.array_equal(det.numpy(), gold["%d_%d_%d" % (A, nc, int(agn))])


def test_sample_instance_embeddings_matches_reference_lines(golden_dir):
    """mot_evaluator.py:1024-1034 exec'd from the reference tree (tests/golden/make_golden_sample.py) vs the oracle restatement."""
    g = np.load(os.path.join(golden_dir, "sample_embed_ref.npz"))
    for tag in ("a", "b"):
        got = uo.sample_instance_embeddings(torch.from_numpy(g["embed_" + tag]), torch.from_numpy(g["boxes_" + tag]))
        assert torch.allclose(got, torch.from_numpy(g["feats_" + tag]), atol=1e-6), tag


def test_label_map_matches_reference_function(golden_dir):
    """`get_label_map` + the driver's 1/8 bilinear down-sampling (unicorn_sot.py:52-53,128-139): the golden maps come from exec-ing
    the reference function (tests/golden/make_golden_labelmap.py) on half-integer, clipped, empty and inverted boxes."""
    g = np.load(os.path.join(golden_dir, "labelmap_ref.npz"))
    for tag in ("a", "b"):
        H, W = (int(v) for v in g["hw_" + tag])
        for b, ref in zip(g["boxes"], g["lbs_" + tag]):
            got = uo.label_map_s8(torch.from_numpy(b), H, W)[0].numpy()
            assert np.array_equal(got, ref), (tag, b)


def test_oracle_loops_match_the_reference_evaluator_methods(golden_dir):
    """The per-frame ORACLE loops the GPU tests lean on (uo.mot_whole -> uo.postprocess -> interaction / upsample / instance embeddings -> assoc_oracle.qd_match, and
    -> bytetrack_oracle.byte_update) against what the reference's OWN evaluator METHODS produced end to end (tests/golden/make_golden_drivers.py: `MOTEvaluator.evaluate_omni`
    and `MOTEvaluator.evaluate`, unmodified, reference model + reference trackers): the tracker outputs per frame and the rows of the MOT result files.  CPU, large model at
    800 x 1280 (~40 s)."""
    import copy
    import assoc_oracle as ao
    import bytetrack_oracle as bo
    from planted import confident_head
    torch.set_num_threads(8)
    exp, H, W = "unicorn_track_large_mot_challenge", 800, 1280
    cfg = uo.CONFIGS[exp]
    P = confident_head(synth.synth_state_dict(cfg), 0.0, 0.0, 2.0)
    # ---- evaluate_omni (mot_evaluator.py:925-1105)
    g = np.load(os.path.join(golden_dir, "driver_omni_%s_%dx%d.npz" % (exp, H, W)))
    n, seed, conf = int(g["nframes"][0]), int(g["seed"][0]), float(g["confthre"][0])
    frames, _ = synth.synth_clip(H, W, n + 1, seed=seed)
    img_h, img_w = (int(v) for v in g["img_hw"])
    scale = min(H / float(img_h), W / float(img_w))
    st, pre, rows = ao.QDState(), None, []                               # QuasiDenseEmbedTracker() defaults
    for t in range(1, n + 1):
        with torch.no_grad():
            o, d, _ = uo.mot_whole(P, cfg, frames[t])
            det = uo.postprocess(o.clone(), 1, conf, 0.7)[0]
            bb, sc = det[:, :4], det[:, 4:5] * det[:, 5:6]
            keep = sc[:, 0] > 0.1
            bb, sc = bb[keep], sc[keep]
            if t == 1:
                pre = copy.deepcopy(d)
            _, fo = uo.forward_interaction(P, pre, d)
            e = uo.forward_upsample(P, fo)
            pre = copy.deepcopy(d)
            emb = uo.sample_instance_embeddings(e, bb)
        b_o, _, ids_o, _ = ao.qd_match(st, torch.cat((bb / scale, sc), 1), torch.ones((bb.shape[0],)), emb, t)
        b_ref, i_ref = torch.from_numpy(g["match_bboxes_%d" % t]), torch.from_numpy(g["match_ids_%d" % t])
        assert b_o.shape == b_ref.shape and (torch.as_tensor(b_o) - b_ref).abs().max() < 1e-3 and torch.equal(torch.as_tensor(ids_o).long(), i_ref.long()), t
        ids_o = torch.as_tensor(ids_o).long()
        v = ids_o > -1
        ob, oi = torch.as_tensor(b_o)[v], ids_o[v]
        order = torch.argsort(oi)
        for i in order.tolist():
            x1, y1, x2, y2, s_ = [float(x) for x in ob[i]]
            w, h = x2 - x1, y2 - y1
            if w * h > 10 and not (w / h > 1.6):
                rows.append([t, int(oi[i]) + 1, round(x1, 1), round(y1, 1), round(w, 1), round(h, 1), round(s_, 2)])
    rr = np.array(rows, dtype=np.float64).reshape(-1, 7)
    assert rr.shape == g["rows"].shape and np.array_equal(rr[:, :2], g["rows"][:, :2]) and np.abs(rr[:, 2:] - g["rows"][:, 2:]).max() <= 0.11
    # ---- evaluate (mot_evaluator.py:100-240): ByteTrack
    g = np.load(os.path.join(golden_dir, "driver_byte_%s_%dx%d.npz" % (exp, H, W)))
    n, seed = int(g["nframes"][0]), int(g["seed"][0])
    base, _ = synth.synth_clip(H, W, 2, seed=seed)
    frames = [base[0]] + [torch.roll(base[1], shifts=max(t - 2, 0), dims=3).contiguous() for t in range(1, n + 1)]
    info = (int(g["img_hw"][0]), int(g["img_hw"][1]))
    bs_ = bo.ByteState(track_thresh=float(g["track_thresh"][0]), track_buffer=30, match_thresh=0.9, mot20=False, frame_rate=30)
    for t in range(1, n + 1):
        with torch.no_grad():
            o, _, _ = uo.mot_whole(P, cfg, frames[t])
            det = uo.postprocess(o.clone(), 1, float(g["confthre"][0]), 0.7)[0]
        tr = bo.byte_update(bs_, det.numpy(), info, (H, W))
        got = np.array([[*x.tlwh, x.track_id, x.score] for x in tr], dtype=np.float64).reshape(-1, 6)
        ref = g["tracks_%d" % t]
        assert got.shape == ref.shape, (t, got.shape, ref.shape)
        got, ref = got[np.argsort(got[:, 4], kind="stable")], ref[np.argsort(ref[:, 4], kind="stable")]
        assert np.array_equal(got[:, 4], ref[:, 4]) and np.abs(got[:, :4] - ref[:, :4]).max() < 1e-3 and np.abs(got[:, 5] - ref[:, 5]).max() < 1e-6, t


@pytest.mark.parametrize("exp", ["unicorn_track_tiny", "unicorn_track_large"])
def test_oracle_sot_driver_matches_the_reference_driver_class(exp, golden_dir):
    """uo.sot_init / sot_step / postprocess / sot_pick_box (the oracle's restatement of unicorn_sot.py:39-108) against what the reference's OWN `UnicornSOTTrack` class
    produced end to end (tests/golden/driver_sot_*_800x1280.npz: raw uint8 frames at the network size, planted scores, ~21000 NMS candidates): the first
    detections per frame and the integer `target_bbox` states.  CPU, tiny (3 frames) and large (2 frames)."""
    from planted import confident_head
    torch.set_num_threads(8)
    H, W = 800, 1280
    g = np.load(os.path.join(golden_dir, "driver_sot_%s_%dx%d.npz" % (exp, H, W)))
    cfg = uo.CONFIGS[exp]
    P = confident_head(synth.synth_state_dict(cfg))
    n, seed = int(g["nframes"][0]), int(g["seed"][0])
    rng = np.random.default_rng(seed)                                  # make_golden_drivers.py:driver_clip; PreprocessorX: RGB -> BGR, r = 1
    base = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    frames = []
    for t in range(n + 1):
        f = np.roll(base, (3 * t, 5 * t), (0, 1)).astype(np.int16) + rng.integers(0, 8, (H, W, 3), dtype=np.int16)
        f = np.clip(f, 0, 255).astype(np.uint8)[:, :, ::-1]
        frames.append(torch.from_numpy(np.ascontiguousarray(f.transpose(2, 0, 1))).float()[None])
    ib = torch.tensor([float(v) for v in g["init_bbox"]])
    box = torch.cat((ib[:2], ib[:2] + ib[2:]))
    with torch.no_grad():
        st = uo.sot_init(P, cfg, frames[0], box)
        for t in range(1, n + 1):
            r = uo.sot_step(P, cfg, st, frames[t])
            det = uo.postprocess(r["head"].clone(), 1, 0.001, 0.65)[0]
            ref = torch.from_numpy(g["det_%d" % t])
            assert abs(det.shape[0] - int(g["n_det_%d" % t][0])) == 0, (t, det.shape[0], int(g["n_det_%d" % t][0]))
            d = det[:ref.shape[0]].clone()
            d[:, 0:4:2] = d[:, 0:4:2].clamp(min=0, max=W)
            d[:, 1:4:2] = d[:, 1:4:2].clamp(min=0, max=H)
            assert (d[:, :4] - ref[:, :4]).abs().max() < 1e-3 and (d[:, 4:6] - ref[:, 4:6]).abs().max() < 1e-6, t
            assert uo.sot_pick_box(det, H, W) == [int(v) for v in g["target_bbox_%d" % t]], t


def test_oracle_vos_driver_matches_the_reference_driver_class(golden_dir):
    """uo.vos_track_init / vos_track_frame (the oracle's restatement of unicorn_vos.py:43-200 incl. the soft aggregation) against the id maps the reference's OWN
    `UnicornVOSTrack` class produced (tests/golden/driver_vos_unicorn_track_tiny_mask_800x1280.npz: K = 3 objects, two frames, biases planted at -4.2 so that a few
    hundred anchors per object pass the driver's confthre = 0.001).  CPU."""
    from planted import confident_head
    torch.set_num_threads(8)
    exp, H, W = "unicorn_track_tiny_mask", 800, 1280
    g = np.load(os.path.join(golden_dir, "driver_vos_%s_%dx%d.npz" % (exp, H, W)))
    cfg = uo.CONFIGS[exp]
    P = confident_head(synth.synth_state_dict(cfg), -4.2, -4.2)
    n, seed = int(g["nframes"][0]), int(g["seed"][0])
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    frames = []
    for t in range(n + 1):
        f = np.roll(base, (3 * t, 5 * t), (0, 1)).astype(np.int16) + rng.integers(0, 8, (H, W, 3), dtype=np.int16)
        f = np.clip(f, 0, 255).astype(np.uint8)[:, :, ::-1]
        frames.append(torch.from_numpy(np.ascontiguousarray(f.transpose(2, 0, 1))).float()[None])
    boxes = {}
    for i, k in enumerate(("1", "2", "3")):
        x, y, w, h = [float(v) for v in g["boxes"][i]]
        boxes[k] = torch.tensor([x, y, x + w, y + h])
    with torch.no_grad():
        st = uo.vos_track_init(P, cfg, frames[0], boxes, (H, W), 1.0)
        nrun = n if os.environ.get("UNI_SLOW_TESTS") else 1            # a frame costs ~90 s of CPU (hundreds of 800 x 1280 CondInst masks per object); both frames passed in round 6
        for t in range(1, nrun + 1):
            seg = uo.vos_track_frame(P, cfg, st, frames[t], {}, 1.0)
            ref = g["seg_%d" % t]
            assert seg.shape == ref.shape and float((seg == ref).mean()) > 0.99999, (t, float((seg == ref).mean()))
