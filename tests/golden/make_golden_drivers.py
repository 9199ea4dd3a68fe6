"""Golden sequences from the REAL reference DRIVER classes executed end to end (run in the build container only).

    python tests/golden/make_golden_drivers.py [sot] [omni] [byte] [vos]

(second half, `run_omni_evaluator`: the reference's `MOTEvaluator.evaluate_omni` method itself on a stand-in dataloader -> the MOT result file it writes)

`external/lib/test/tracker/unicorn_sot.py:UnicornSOTTrack` -- the class `tools/test.py unicorn_sot ...` instantiates -- is imported from
/root/reference UNMODIFIED and driven exactly like `lib/test/evaluation/tracker.py:138-198` drives it (`initialize(image, info)`, then `track(image)`
per frame) with the reference's own model behind it (exp.get_model(), reference nn.Modules on the CPU, synthetic weights loaded through the
driver's own `torch.load(params.checkpoint)["model"]` path).  What has to be substituted for the class to run in this container:
  * `cv2` (absent offline): `cvtColor(RGB2BGR)` = channel flip; `resize` is only ever asked for the IDENTITY size here (the clip is generated at the
    network input size, r = 1), so no third-party interpolation arithmetic enters the golden;
  * `.cuda()` / `device="cuda"` -> CPU (oracle/ref_bootstrap.py + Tensor.cuda / Module.cuda / Tensor.to("cuda") no-ops);
  * the fp16 casts of the correlation (`.half()`, unicorn_sot.py:95-97) are dropped: parity is defined against the fp32 statement;
  * `np.int` (removed from numpy) -> int.
Detector-like scores are planted (tests/planted.py:confident_head) so that `postprocess` returns boxes (synthetic scores ~1e-4 < confthre 0.001).
Stored per frame: the driver's float detections (first max_inst rows of get_det_results after the clamp) and its integer `target_bbox` state.
The GPU test drives `unicorn_amd.tracker.UnicornSOTTrack` + the HIP model with the same frames and holds it to these rows."""
import os
import sys
import types

import numpy as np
import tabulate  # noqa: F401  (the real package, before ref_bootstrap would stub it: torch._dynamo probes its __spec__)
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
sys.path.insert(0, os.path.join(HERE, ".."))
import ref_bootstrap as rb  # noqa: E402
import synth  # noqa: E402
import unicorn_oracle as uo  # noqa: E402
from planted import confident_head  # noqa: E402


def driver_clip(H, W, n, seed):
    """n + 1 uint8 RGB frames (H, W, 3) at the network input size: a random texture rolled by (3t, 5t) px + small noise"""
    g = np.random.default_rng(seed)
    base = g.integers(0, 256, (H, W, 3), dtype=np.uint8)
    out = []
    for t in range(n + 1):
        f = np.roll(base, (3 * t, 5 * t), (0, 1)).astype(np.int16) + g.integers(0, 8, (H, W, 3), dtype=np.int16)
        out.append(np.clip(f, 0, 255).astype(np.uint8))
    return out


def install_driver_patches():
    rb.boot()
    cv2 = sys.modules["cv2"]
    cv2.COLOR_RGB2BGR, cv2.INTER_LINEAR = 4, 1

    def cvtColor(img, code):
        assert code == cv2.COLOR_RGB2BGR
        return np.ascontiguousarray(img[:, :, ::-1])

    def resize(img, size, interpolation=None):
        assert (img.shape[1], img.shape[0]) == tuple(size), "only the identity resize is allowed here (cv2 is absent: its arithmetic must not enter the golden)"
        return img
    cv2.cvtColor, cv2.resize = cvtColor, resize
    # a host <-> device transfer is a COPY: the reference relies on it (MOTEvaluator.convert_to_coco_format rescales `output.cpu()[:, :4]` in place,
    # mot_evaluator.py:623-632, before the same `outputs` go to the tracker) -- no-op transfers would alias and corrupt the boxes on a CPU-only run
    torch.Tensor.cuda = lambda self, *a, **k: self.clone()
    torch.Tensor.cpu = lambda self, *a, **k: self.clone()
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.Tensor.half = lambda self: self.float()          # the fp16 casts of the propagation are dropped (fp32 statement)
    _to = torch.Tensor.to

    def to(self, *a, **k):
        moved = any(isinstance(v, str) and v.startswith("cuda") for v in a) or (isinstance(k.get("device"), str) and k["device"].startswith("cuda"))
        a = tuple("cpu" if isinstance(v, str) and v.startswith("cuda") else v for v in a)
        if isinstance(k.get("device"), str) and k["device"].startswith("cuda"):
            k["device"] = "cpu"
        r = _to(self, *a, **k)
        return r.clone() if moved and r is self else r
    torch.Tensor.to = to
    ext = os.path.join(rb.REF_ROOT, "external")
    if ext not in sys.path:
        sys.path.insert(0, ext)


def run_sot_driver(exp_name, H, W, nframes, seed):
    install_driver_patches()
    from lib.test.tracker.unicorn_sot import UnicornSOTTrack          # the reference class, unmodified
    cfg = uo.CONFIGS[exp_name]
    P = confident_head(synth.synth_state_dict(cfg))
    ck = "/tmp/_driver_ckpt_%s.pth" % exp_name
    torch.save({"model": P}, ck)
    cwd = os.getcwd()
    os.chdir(rb.REF_ROOT)                                              # get_exp("exps/default/<name>") is relative to the repo root
    try:
        trk = UnicornSOTTrack(types.SimpleNamespace(exp_name=exp_name, checkpoint=ck), "synthetic")
    finally:
        os.chdir(cwd)
    assert tuple(trk.input_size) == (H, W), trk.input_size
    trk.device = "cpu"
    frames = driver_clip(H, W, nframes, seed)
    init = [W * 0.25, H * 0.25, W * 0.25, H * 0.25]                    # xywh
    dets = []
    inner = trk.get_det_results

    def tap(cur):
        o = inner(cur)
        dets.append(None if o is None else o.detach().clone())
        return o
    trk.get_det_results = tap
    trk.initialize(frames[0], {"init_bbox": list(init)})
    out = {"init_bbox": np.array(init, dtype=np.float64), "seed": np.array([seed]), "size": np.array([H, W]), "nframes": np.array([nframes])}
    for t in range(1, nframes + 1):
        res = trk.track(frames[t])
        d = dets[-1]
        out["n_det_%d" % t] = np.array([0 if d is None else d.shape[0]])
        if d is not None:
            d = d.clone()
            d[:, 0:4:2] = d[:, 0:4:2].clamp(min=0, max=W)              # track() clamps a copy that get_det_results returned (:64-65, in place on the same tensor)
            d[:, 1:4:2] = d[:, 1:4:2].clamp(min=0, max=H)
            out["det_%d" % t] = d[:trk.max_inst].numpy().astype(np.float32)
        out["target_bbox_%d" % t] = np.array(res["target_bbox"], dtype=np.int64)
        print(exp_name, "frame", t, "detections", out["n_det_%d" % t][0], "state", res["target_bbox"], flush=True)
    np.savez_compressed(os.path.join(HERE, "driver_sot_%s_%dx%d.npz" % (exp_name, H, W)), **out)


def run_omni_evaluator(exp_name, H, W, nframes, seed, ncand=300):
    """`MOTEvaluator.evaluate_omni` (unicorn/evaluators/mot_evaluator.py:925-1105) -- the method `tools/track_omni.py` calls -- executed UNMODIFIED on a
    stand-in dataloader of `nframes` frames of one video, with the reference's own model and the reference's own `QuasiDenseEmbedTracker()`; the MOT result
    file it writes (`write_results`, :49-58) is the golden.  Substituted: `torch.cuda.FloatTensor` / `synchronize` on the CPU, `evaluate_prediction`
    (COCOeval glue) -> None.  `confthre` is the evaluator's constructor argument (tools/track_omni.py --conf): set between two neighbouring scores of
    the first frame so that ~`ncand` candidates pass."""
    install_driver_patches()
    import tempfile
    import torch.nn.functional as F  # noqa: F401
    torch.cuda.FloatTensor = torch.FloatTensor
    torch.cuda.synchronize = lambda *a, **k: None
    for absent in ("mmcv", "scalabel", "scalabel.label", "scalabel.label.io", "scalabel.label.transforms", "scalabel.label.typing", "scalabel.eval",
                   "scalabel.eval.mot", "scalabel.eval.detect", "scalabel.eval.ins_seg", "scalabel.eval.mots", "scalabel.label.to_coco"):
        if absent not in sys.modules:          # imported by unicorn/evaluators/__init__.py through the BDD100K evaluator only (out of scope); never called here
            sys.modules[absent] = rb._Anything.__new__(rb._Anything) if False else types.ModuleType(absent)
    from unicorn.evaluators.mot_evaluator import MOTEvaluator
    cfg = uo.CONFIGS[exp_name]
    P = confident_head(synth.synth_state_dict(cfg), 0.0, 0.0, 2.0)
    model, _ = rb.build_reference_model(exp_name)
    missing, unexpected = model.load_state_dict(P, strict=False)
    assert not unexpected and all("mask_head" in m for m in missing)
    frames, _ = synth.synth_clip(H, W, nframes + 1, seed=seed)
    with torch.no_grad():
        o, _ = model(frames[1], mode="whole")
    sc = (o[0, :, 4] * o[0, :, 5]).sort(descending=True)[0]
    confthre = float((sc[ncand - 1] + sc[ncand]) / 2)
    img_h, img_w = int(H * 1.35), int(W * 1.35)

    class Loader(list):
        dataset = types.SimpleNamespace(class_ids=[1])
    loader = Loader()
    for t in range(1, nframes + 1):
        info = (torch.tensor([img_h]), torch.tensor([img_w]), torch.tensor([t]), torch.tensor([1]), ["SYN-01/img1/%06d.jpg" % t])
        loader.append((frames[t], None, info, torch.tensor([t])))
    ev = MOTEvaluator(types.SimpleNamespace(min_box_area=10), loader, (H, W), confthre, 0.7, 1)      # (--min_box_area 10: the synthetic heads regress small boxes)
    ev.evaluate_prediction = lambda data_list, statistics: None
    out_dir = tempfile.mkdtemp()
    # observe (not alter) what the reference tracker returns per frame: the result file keeps only tracks that pass the area / aspect filter (:1069-1072)
    import unicorn.tracker.quasi_dense_embed_tracker as qd
    seen, orig_match = [], qd.QuasiDenseEmbedTracker.match

    def tap(self, bboxes, labels, track_feats, frame_id, *a, **k):
        r = orig_match(self, bboxes, labels, track_feats, frame_id, *a, **k)
        seen.append((int(frame_id), r[0].detach().clone(), r[2].detach().clone()))
        return r
    qd.QuasiDenseEmbedTracker.match = tap
    try:
        ev.evaluate_omni(model, result_folder=out_dir)
    finally:
        qd.QuasiDenseEmbedTracker.match = orig_match
    rows = np.loadtxt(os.path.join(out_dir, "SYN-01.txt"), delimiter=",", ndmin=2)[:, :7]      # frame, id, x1, y1, w, h, score (rounded by write_results)
    print(exp_name, "evaluate_omni:", {int(f): int((rows[:, 0] == f).sum()) for f in np.unique(rows[:, 0])}, "confthre", confthre,
          "match outputs per frame:", [(f, int(b.shape[0]), int((i > -1).sum())) for f, b, i in seen], flush=True)
    out = dict(rows=rows.astype(np.float64), confthre=np.array([confthre]), seed=np.array([seed]), nframes=np.array([nframes]),
               img_hw=np.array([img_h, img_w]), size=np.array([H, W]))
    for f, b, i in seen:
        out["match_bboxes_%d" % f] = b.numpy().astype(np.float32)
        out["match_ids_%d" % f] = i.numpy().astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "driver_omni_%s_%dx%d.npz" % (exp_name, H, W)), **out)


def run_vos_driver(exp_name, H, W, nframes, seed):
    """`external/lib/test/tracker/unicorn_vos.py:UnicornVOSTrack` (the class `tools/test.py unicorn_vos ...` instantiates), unmodified, K = 3 objects given in
    the first frame: initialize, then track per frame -> the (H, W) uint8 id map of every frame (soft aggregation, :99-120) and the per-object integer box states.
    Prediction biases are planted at -4.2 (synthetic: -4.5) so that a few hundred of the 21000 anchors pass the driver's confthre = 0.001 per object and
    `postprocess_inst` has CondInst masks to compute (at -4.5 nothing passes; with detector-like scores all 21000 would, 86 GB of masks)."""
    install_driver_patches()
    from lib.test.tracker.unicorn_vos import UnicornVOSTrack
    cfg = uo.CONFIGS[exp_name]
    P = confident_head(synth.synth_state_dict(cfg), -4.2, -4.2)
    ck = "/tmp/_driver_ckpt_%s.pth" % exp_name
    ref_model, _ = rb.build_reference_model(exp_name)                  # the VOS driver loads strictly: a released checkpoint also carries the mask head's two buffers
    bufs = {k: v.clone() for k, v in ref_model.state_dict().items() if k.startswith("head.mask_head.")}
    del ref_model
    torch.save({"model": dict(P, **bufs)}, ck)
    cwd = os.getcwd()
    os.chdir(rb.REF_ROOT)
    try:
        trk = UnicornVOSTrack(types.SimpleNamespace(exp_name=exp_name, checkpoint=ck), "synthetic")
    finally:
        os.chdir(cwd)
    assert tuple(trk.input_size) == (H, W), trk.input_size
    trk.device = "cpu"
    frames = driver_clip(H, W, nframes, seed)
    boxes = {"1": [W * 0.25, H * 0.25, W * 0.25, H * 0.25], "2": [W * 0.55, H * 0.1, W * 0.35, H * 0.35], "3": [W * 0.1, H * 0.55, W * 0.3, H * 0.4]}      # xywh
    trk.initialize(frames[0], {"init_object_ids": list(boxes), "sequence_object_ids": list(boxes), "init_bbox": {k: list(v) for k, v in boxes.items()}})
    out = {"seed": np.array([seed]), "size": np.array([H, W]), "nframes": np.array([nframes]), "boxes": np.array([boxes[k] for k in boxes], dtype=np.float64)}
    for t in range(1, nframes + 1):
        seg = trk.track(frames[t], {})["segmentation"]
        out["seg_%d" % t] = seg.astype(np.uint8)
        out["states_%d" % t] = np.array([trk.state_pre_dict[k] for k in boxes], dtype=np.float64)
        print(exp_name, "vos frame", t, "pixels per id", {int(i): int((seg == i).sum()) for i in np.unique(seg)}, "states", out["states_%d" % t].tolist(), flush=True)
    np.savez_compressed(os.path.join(HERE, "driver_vos_%s_%dx%d.npz" % (exp_name, H, W)), **out)


def byte_clip(H, W, nframes, seed):
    """a nearly static video: randomly initialised heads regress boxes that do not persist under the (3t, 5t) px motion of synth_clip, and an IoU tracker then
    has nothing to associate -- frames 1 and 2 are the same image (every track re-associates: Kalman update, confirmation of the tracks born in frame 2), frame t > 2 is shifted by (t - 2) px along x"""
    base, _ = synth.synth_clip(H, W, 2, seed=seed)
    return [base[0]] + [torch.roll(base[1], shifts=max(t - 2, 0), dims=3).contiguous() for t in range(1, nframes + 1)]      # shifts 0, 0, 1, 2, ...


def run_byte_evaluator(exp_name, H, W, nframes, seed, ncand=150, ntrk=120):
    """`MOTEvaluator.evaluate` (mot_evaluator.py:100-240) -- the method `tools/track.py` calls -- executed UNMODIFIED like run_omni_evaluator, with the reference's
    own `BYTETracker` (byte_tracker.py / matching.py / kalman_filter.py; its absent third-party solvers `lap.lapjv` / `cython_bbox.bbox_overlaps` are the
    restatements of oracle/bytetrack_oracle.py -- declared unpinned).  `track_thresh` / `confthre` are user arguments (tools/track.py:100-108), set between
    neighbouring scores of the first frame."""
    install_driver_patches()
    import tempfile
    torch.cuda.FloatTensor = torch.FloatTensor
    torch.cuda.synchronize = lambda *a, **k: None
    for absent in ("mmcv", "scalabel", "scalabel.label", "scalabel.label.io", "scalabel.label.transforms", "scalabel.label.typing", "scalabel.eval",
                   "scalabel.eval.mot", "scalabel.eval.detect", "scalabel.eval.ins_seg", "scalabel.eval.mots", "scalabel.label.to_coco"):
        if absent not in sys.modules:
            sys.modules[absent] = types.ModuleType(absent)
    from unicorn.evaluators.mot_evaluator import MOTEvaluator
    import unicorn.tracker.byte_tracker as bt
    cfg = uo.CONFIGS[exp_name]
    P = confident_head(synth.synth_state_dict(cfg), 0.0, 0.0, 2.0)
    model, _ = rb.build_reference_model(exp_name)
    missing, unexpected = model.load_state_dict(P, strict=False)
    assert not unexpected and all("mask_head" in m for m in missing)
    frames = byte_clip(H, W, nframes, seed)
    with torch.no_grad():
        o, _ = model(frames[1], mode="whole")
    sc = (o[0, :, 4] * o[0, :, 5]).sort(descending=True)[0]
    confthre, track_thresh = float((sc[ncand - 1] + sc[ncand]) / 2), float((sc[ntrk - 1] + sc[ntrk]) / 2)
    img_h, img_w = int(H * 1.35), int(W * 1.35)

    class Loader(list):
        dataset = types.SimpleNamespace(class_ids=[1])
    loader = Loader()
    for t in range(1, nframes + 1):
        info = (torch.tensor([img_h]), torch.tensor([img_w]), torch.tensor([t]), torch.tensor([1]), ["SYN-02/img1/%06d.jpg" % t])
        loader.append((frames[t], None, info, torch.tensor([t])))
    args = types.SimpleNamespace(track_thresh=track_thresh, track_buffer=30, match_thresh=0.9, mot20=False, min_box_area=10)
    ev = MOTEvaluator(args, loader, (H, W), confthre, 0.7, 1)
    ev.evaluate_prediction = lambda data_list, statistics: None
    out_dir = tempfile.mkdtemp()
    seen, orig_update = [], bt.BYTETracker.update

    def tap(self, output_results, img_info, img_size):
        r = orig_update(self, output_results, img_info, img_size)
        seen.append(np.array([[*t_.tlwh, t_.track_id, t_.score] for t_ in r], dtype=np.float64).reshape(-1, 6))
        return r
    bt.BYTETracker.update = tap
    bt.BaseTrack._count = 0
    try:
        ev.evaluate(model, result_folder=out_dir)
    finally:
        bt.BYTETracker.update = orig_update
    rows = np.loadtxt(os.path.join(out_dir, "SYN-02.txt"), delimiter=",", ndmin=2)[:, :7]
    print(exp_name, "evaluate (ByteTrack):", {int(f): int((rows[:, 0] == f).sum()) for f in np.unique(rows[:, 0])}, "confthre", confthre, "track_thresh", track_thresh,
          "tracks returned per frame:", [int(v.shape[0]) for v in seen], flush=True)
    out = dict(rows=rows.astype(np.float64), confthre=np.array([confthre]), track_thresh=np.array([track_thresh]), seed=np.array([seed]), nframes=np.array([nframes]),
               img_hw=np.array([img_h, img_w]), size=np.array([H, W]))
    for t, v in enumerate(seen):
        out["tracks_%d" % (t + 1)] = v
    np.savez_compressed(os.path.join(HERE, "driver_byte_%s_%dx%d.npz" % (exp_name, H, W)), **out)


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["sot", "omni", "byte", "vos"]
    if "vos" in which:
        run_vos_driver("unicorn_track_tiny_mask", 800, 1280, 2, seed=25)
    if "byte" in which:
        run_byte_evaluator("unicorn_track_large_mot_challenge", 800, 1280, 4, seed=24)
    if "sot" in which:
        run_sot_driver("unicorn_track_tiny", 800, 1280, 3, seed=21)
        run_sot_driver("unicorn_track_large", 800, 1280, 2, seed=22)
    if "omni" in which:
        run_omni_evaluator("unicorn_track_large_mot_challenge", 800, 1280, 3, seed=23)
