"""Row N4 (checkpoint tooling): the product-side state spec against the specs dumped from the REAL reference models, and the
load / remap rules of unicorn/utils/checkpoint.py:11-33 and unicorn/exp/unicorn_track.py:163-186.  CPU only."""
import json
import os

import pytest
import torch

from unicorn_amd.utils import checkpoint as ck

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["unicorn_track_tiny", "unicorn_track_tiny_mask", "unicorn_track_large", "unicorn_track_large_mask",
         "unicorn_track_large_mot_challenge"]


@pytest.mark.parametrize("name", NAMES)
def test_state_spec_matches_reference_dump(name):
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "state_spec_%s.json" % name)))
    spec = ck.state_spec(name)
    assert list(spec.keys()) == list(gold.keys()) or set(spec.keys()) == set(gold.keys())
    for k, shp in spec.items():
        assert list(shp) == list(gold[k]), k


def test_filter_and_load_ckpt_rules():
    spec = ck.state_spec("unicorn_track_tiny")
    ckpt = {k: torch.zeros(v) for k, v in spec.items()}
    del ckpt["head.beta_0"]                                          # missing
    ckpt["head.cls_preds.0.weight"] = torch.zeros(80, 256, 1, 1)     # shape mismatch (COCO head)
    ckpt["extra.key"] = torch.zeros(3)                               # unexpected: ignored by load_ckpt
    load, missing, mism = ck.filter_ckpt(spec, ckpt)
    assert missing == ["head.beta_0"] and mism == ["head.cls_preds.0.weight"]
    assert "extra.key" not in load and len(load) == len(spec) - 2

    class Fake:
        name = "unicorn_track_tiny"

        def load_state_dict(self, sd, strict=True):
            self.sd, self.strict = sd, strict

    m = ck.load_ckpt(Fake(), ckpt)
    assert m.strict is False and set(m.sd) == set(load)


def test_remap_coco_pretrain():
    g = torch.Generator().manual_seed(0)
    sd = {"backbone.x": torch.randn(3, generator=g)}
    for k in range(3):
        sd["head.cls_preds.%d.weight" % k] = torch.randn(80, 256, 1, 1, generator=g)
        sd["head.cls_preds.%d.bias" % k] = torch.randn(80, generator=g)
        sd["head.obj_preds.%d.weight" % k] = torch.randn(1, 256, 1, 1, generator=g)
        sd["head.reg_preds.%d.bias" % k] = torch.randn(4, generator=g)
    sd["head.stems.0.conv.weight"] = torch.randn(2, generator=g)
    o8 = ck.remap_coco_pretrain(sd, 8)
    idx = [0, 0, 2, 7, 5, 6, 3, 1]
    assert torch.equal(o8["head.cls_preds.1.weight"], sd["head.cls_preds.1.weight"][idx])
    assert torch.equal(o8["head.cls_preds.2.bias"], sd["head.cls_preds.2.bias"][idx])
    assert torch.equal(o8["head.obj_preds_sot.0.weight"], sd["head.obj_preds.0.weight"]) and "head.obj_preds.0.weight" in o8
    assert torch.equal(o8["head.reg_preds_sot.2.bias"], sd["head.reg_preds.2.bias"])
    assert o8["backbone.x"] is sd["backbone.x"] and o8["head.stems.0.conv.weight"] is sd["head.stems.0.conv.weight"]
    o1 = ck.remap_coco_pretrain(sd, 1, unshared_obj=False, unshared_reg=False)
    assert o1["head.cls_preds.0.bias"].shape == (1,) and "head.obj_preds_sot.0.weight" not in o1
    with pytest.raises(ValueError):
        ck.remap_coco_pretrain(sd, 3)


def test_load_checkpoint_file_unwraps_model_key(tmp_path):
    class Fake:
        def load_state_dict(self, sd, strict=True):
            self.sd = sd
            return "ok"

    p = tmp_path / "latest_ckpt.pth"
    torch.save({"model": {"a": torch.ones(2)}, "start_epoch": 3}, p)
    m = Fake()
    assert ck.load_checkpoint_file(m, str(p)) == "ok" and list(m.sd) == ["a"]


def test_load_state_dict_before_cuda_reports_keys_and_rejects_wrong_shapes():
    """tools/track.py:176-188 order (load_state_dict, then .cuda()): the key report and `strict` must not depend on the order,
    and a checkpoint of another configuration (nc = 1 vs nc = 8) raises like nn.Module does.  No device needed."""
    from unicorn_amd.models import Unicorn
    spec = ck.state_spec("unicorn_track_tiny")
    sd = {k: torch.zeros(v) for k, v in spec.items()}
    del sd["head.beta_1"]
    sd["head.some_new_tensor"] = torch.zeros(3)
    res = Unicorn("unicorn_track_tiny").load_state_dict(sd, strict=False)
    assert list(res.missing_keys) == ["head.beta_1"] and list(res.unexpected_keys) == ["head.some_new_tensor"]
    with pytest.raises(RuntimeError):
        Unicorn("unicorn_track_tiny").load_state_dict(sd, strict=True)
    sd = {k: torch.zeros(v) for k, v in ck.state_spec("unicorn_track_large_mot_challenge").items()}
    with pytest.raises(RuntimeError, match="size mismatch"):
        Unicorn("unicorn_track_large").load_state_dict(sd, strict=False)
