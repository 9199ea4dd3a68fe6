"""tools/validate_checkpoint.py, CPU half: a `{"model": state_dict}` file saved from the REAL reference model
(/root/reference, own initialisation) is loaded the way tools/track.py:186-190 loads a released checkpoint, held to the
learnable-tensor spec, and the oracle is held to the real reference ON THAT CHECKPOINT (the reference tree exists in the build
container only: skipped elsewhere).  The GPU half (HIP path vs oracle + saturation statistics) is tests/test_model_gpu.py."""
import importlib.util
import json
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import ref_bootstrap as rb  # noqa: E402


def _tool():
    spec = importlib.util.spec_from_file_location("validate_checkpoint", os.path.join(ROOT, "tools", "validate_checkpoint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.skipif(not rb.reference_available(), reason="reference tree not present")
def test_real_reference_checkpoint_cpu_only(tmp_path, capsys):
    torch.manual_seed(7)
    model, _ = rb.build_reference_model("unicorn_track_tiny_mask")          # the reference's OWN init (trunc_normal, bias init, MSDA grid)
    path = str(tmp_path / "latest_ckpt.pth")
    torch.save({"model": model.state_dict(), "start_epoch": 1}, path)       # the released file layout (tools/track.py:186-188)
    vc = _tool()
    out = str(tmp_path / "report.json")
    rc = vc.main(["--ckpt", path, "--exp", "unicorn_track_tiny_mask", "--ref", rb.REF_ROOT, "--cpu-only", "--size", "320x320",
                  "--frames", "1", "--json", out, "--threads", "4"])
    capsys.readouterr()
    rep = json.load(open(out))
    assert rc == 0 and rep["pass"], rep
    assert rep["spec"]["ok"] and not rep["spec"]["missing"] and rep["spec"]["unexpected_count"] <= 4      # buffers only (sizes_of_interest, _iter)
    m = rep["oracle_vs_reference"]["per_frame"][0]
    assert rep["oracle_vs_reference"]["pass"] and m["box_iou_min_top500"] > 0.999 and m["embed_cos_min"] > 1 - 1e-4, m
    assert max(m["dyn_params"], m["mask_feats"], m["up_masks"]) < 2e-4, m


def test_spec_failures_exit_nonzero(tmp_path, capsys):
    """a checkpoint of the wrong experiment (or with a missing tensor) must fail before anything runs"""
    from unicorn_amd.utils.checkpoint import state_spec
    spec = state_spec("unicorn_track_tiny")
    sd = {k: torch.zeros(v) for k, v in spec.items()}
    del sd["head.beta_0"]
    sd["bottleneck.0.weight"] = torch.zeros(256, 7, 1, 1)
    path = str(tmp_path / "bad.pth")
    torch.save({"model": sd}, path)
    vc = _tool()
    rc = vc.main(["--ckpt", path, "--exp", "unicorn_track_tiny", "--cpu-only", "--size", "320x320"])
    rep = json.loads(capsys.readouterr().out)
    assert rc == 1 and not rep["pass"]
    assert rep["spec"]["missing"] == ["head.beta_0"] and rep["spec"]["shape_mismatch"] == ["bottleneck.0.weight"]


def test_images_option_letterboxes_npy_frames(tmp_path, capsys):
    """--images: HxWx3 uint8 RGB .npy frames go through the PreprocessorX.process restatement (RGB->BGR, resize by r, pad 114) and the
    init box is scaled by r; --cpu-only without --ref = spec check + finite oracle outputs"""
    import numpy as np
    import synth
    import unicorn_oracle as uo
    P = synth.synth_state_dict(uo.CONFIGS["unicorn_track_tiny"])
    ck = str(tmp_path / "c.pth")
    torch.save({"model": P}, ck)
    g = np.random.default_rng(0)
    files = []
    for i in range(2):
        f = str(tmp_path / ("f%d.npy" % i))
        np.save(f, g.integers(0, 256, (240, 427, 3), dtype=np.uint8))
        files.append(f)
    vc = _tool()
    rc = vc.main(["--ckpt", ck, "--exp", "unicorn_track_tiny", "--cpu-only", "--size", "320x512", "--frames", "1", "--images", *files,
                  "--box", "100,60,220,180", "--threads", "4"])
    rep = json.loads(capsys.readouterr().out)
    assert rc == 0 and rep["pass"] and rep["clip"] == "images" and rep["oracle_finite"]
    args = vc.parse(["--ckpt", ck, "--exp", "unicorn_track_tiny", "--images", *files, "--frames", "1"])
    frames, box = vc.make_clip(args, 320, 512)
    r = min(320 / 240, 512 / 427)
    assert frames[0].shape == (1, 3, 320, 512) and float(frames[0][0, :, int(240 * r) + 1:].min()) == 114.0      # pad rows
    assert torch.allclose(box, torch.tensor([427 / 4, 240 / 4, 3 * 427 / 4, 3 * 240 / 4]) * r)      # the centred default box
