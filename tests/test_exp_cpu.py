"""The exp.get_model() seam (unicorn/exp/build.py:35-50, exp/unicorn_track.py:115, exp/unicorn_track_mask.py:49) on the
HIP package: the mixin reads a REFERENCE-shaped Exp's attributes; no device needed (the model binds to the GPU at .cuda())."""
import pytest

from unicorn_amd.exp import Exp, HipModelMixin, get_exp, model_cfg_from_exp, patch_exp
from unicorn_amd.models import MODEL_CONFIGS, Unicorn


class RefExpTrack:
    """attributes of unicorn.exp.ExpTrack.__init__ that shape the inference model (exp/unicorn_track.py:31-113)"""

    def __init__(self):
        self.exp_name = "ref"
        self.num_classes = 8
        self.backbone_name = "convnext"
        self.in_channels = [192, 384, 768]
        self.embed_dim = 128
        self.interact_mode = "deform"
        self.use_attention = True
        self.n_layer_att = 3
        self.test_size = (800, 1280)

    def get_model(self, load_pretrain=True):
        raise AssertionError("the reference get_model must be overridden")


class RefLargeMotMask(RefExpTrack):      # exps/default/unicorn_track_large_mot_challenge_mask.py + ExpTrackMask attributes
    def __init__(self):
        super().__init__()
        self.backbone_name = "convnext_large"
        self.in_channels = [384, 768, 1536]
        self.num_classes = 1
        self.use_raft, self.d_rate, self.ctrl_loc = True, 2, "reg"


def test_mixin_in_front_of_a_reference_exp_class():
    class HipExp(HipModelMixin, RefExpTrack):
        pass
    m = HipExp().get_model(load_pretrain=False)
    assert isinstance(m, Unicorn) and m.dims == (96, 192, 384, 768) and m.depths == (3, 3, 9, 3)
    assert m.num_classes == 8 and not m.mask and m.n_layer_att == 3 and m.precision == "f16x2"


def test_patch_exp_object_and_mask_variant():
    exp = patch_exp(RefLargeMotMask(), precision="bf16")
    m = exp.get_model()
    assert exp.get_model() is m                                   # cached like the reference (`self.model`)
    assert m.dims == (192, 384, 768, 1536) and m.depths == (3, 3, 27, 3) and m.num_classes == 1 and m.mask
    assert m.up_rate == 4 and m.precision == "bf16" and hasattr(m.head, "mask_head")


@pytest.mark.parametrize("name", sorted(MODEL_CONFIGS))
def test_get_exp_by_name_matches_model_configs(name):
    exp = get_exp(exp_name=name)
    assert isinstance(exp, Exp) and exp.test_size == (800, 1280)
    cfg = model_cfg_from_exp(exp)
    ref = MODEL_CONFIGS[name]
    assert tuple(cfg["dims"]) == tuple(ref["dims"]) and tuple(cfg["depths"]) == tuple(ref["depths"])
    assert cfg["num_classes"] == ref["num_classes"] and cfg["mask"] == ref["mask"]
    m = get_exp(exp_file="exps/default/%s.py" % name).get_model()       # file path form (reference absent -> resolved by name)
    assert m.num_classes == ref["num_classes"] and m.mask == ref["mask"]


def test_unsupported_exps_fail_loudly():
    e = RefExpTrack()
    e.backbone_name = "resnet50"
    with pytest.raises(ValueError):
        model_cfg_from_exp(e)
    e = RefExpTrack()
    e.interact_mode = "full"
    with pytest.raises(ValueError):
        model_cfg_from_exp(e)
    with pytest.raises(ValueError):
        get_exp(exp_name="unicorn_track_r50")
