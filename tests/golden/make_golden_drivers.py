"""Golden sequences from the REAL reference DRIVER classes executed end to end (run in the build container only).

    python tests/golden/make_golden_drivers.py

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
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.Tensor.half = lambda self: self.float()          # the fp16 casts of the propagation are dropped (fp32 statement)
    _to = torch.Tensor.to

    def to(self, *a, **k):
        a = tuple("cpu" if isinstance(v, str) and v.startswith("cuda") else v for v in a)
        if isinstance(k.get("device"), str) and k["device"].startswith("cuda"):
            k["device"] = "cpu"
        return _to(self, *a, **k)
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


if __name__ == "__main__":
    torch.set_num_threads(8)
    run_sot_driver("unicorn_track_tiny", 800, 1280, 3, seed=21)
    run_sot_driver("unicorn_track_large", 800, 1280, 2, seed=22)
