"""The `exp.get_model()` seam of the reference (unicorn/exp/build.py:35-50 -> exp/unicorn_track.py:115,
exp/unicorn_track_mask.py:49) on the HIP path.

Two ways in:
  * `HipModelMixin`: put it in front of a reference Exp class (`class Exp(HipModelMixin, unicorn.exp.ExpTrack)`), or call
    `patch_exp(exp)` on an already built exp object: `exp.get_model()` then returns `unicorn_amd.models.Unicorn` configured
    from the exp's own attributes (backbone_name, in_channels, num_classes, n_layer_att, embed_dim, d_rate / use_raft for the
    mask variants) - tools/track.py, tools/demo.py and the lib/test trackers call nothing else on the model than the surface
    that class mirrors.
  * `get_exp(exp_file=None, exp_name=None)`: same signature as unicorn.exp.get_exp.  With the reference importable it loads
    the reference exp file and patches it; without it (the GPU box has no /root/reference) the shipped exps/default names
    resolve to stand-alone `Exp` objects carrying the inference-relevant attributes of those files.
"""
import importlib
import os
import sys

# ConvNeXt variants of backbone/convnext.py:198-211 reachable through YOLOPAFPNNEW(backbone_name=...)
_CONVNEXT = {
    "convnext": dict(dims=(96, 192, 384, 768), depths=(3, 3, 9, 3)),            # exp/unicorn_track.py:42 default = tiny
    "convnext_tiny": dict(dims=(96, 192, 384, 768), depths=(3, 3, 9, 3)),
    "convnext_large": dict(dims=(192, 384, 768, 1536), depths=(3, 3, 27, 3)),
}

# exps/default/*.py (every ConvNeXt tracking exp the reference ships): (backbone_name, num_classes, mask, test_size)
_DEFAULT_EXPS = {
    "unicorn_track_tiny": ("convnext_tiny", 8, False, (800, 1280)),
    "unicorn_track_tiny_mask": ("convnext_tiny", 8, True, (800, 1280)),
    "unicorn_track_tiny_rt": ("convnext_tiny", 8, False, (640, 1024)),             # exps/default/unicorn_track_tiny_rt.py:16-17
    "unicorn_track_tiny_rt_mask": ("convnext_tiny", 8, True, (640, 1024)),
    "unicorn_track_tiny_mot_only": ("convnext_tiny", 8, False, (800, 1280)),       # ablations: training-side switches only
    "unicorn_track_tiny_sot_only": ("convnext_tiny", 8, False, (800, 1280)),
    "unicorn_track_tiny_mots_only": ("convnext_tiny", 8, True, (800, 1280)),
    "unicorn_track_tiny_vos_only": ("convnext_tiny", 8, True, (800, 1280)),
    "unicorn_track_large": ("convnext_large", 8, False, (800, 1280)),
    "unicorn_track_large_mask": ("convnext_large", 8, True, (800, 1280)),
    "unicorn_track_large_mot_challenge": ("convnext_large", 1, False, (800, 1280)),
    "unicorn_track_large_mot_challenge_mask": ("convnext_large", 1, True, (800, 1280)),
}


def model_cfg_from_exp(exp):
    """inference-relevant attributes of a reference Exp (exp/unicorn_track.py:31-113, unicorn_track_mask.py:31-47) -> Unicorn cfg"""
    name = getattr(exp, "backbone_name", "convnext")
    if name not in _CONVNEXT:
        raise ValueError("unicorn_amd implements the ConvNeXt backbones (convnext_tiny / convnext_large); got backbone_name=%r" % name)
    if getattr(exp, "interact_mode", "deform") != "deform":
        raise ValueError("unicorn_amd implements interact_mode='deform' only (got %r)" % exp.interact_mode)
    cfg = dict(_CONVNEXT[name])
    inch = tuple(getattr(exp, "in_channels", cfg["dims"][1:]))
    if inch != tuple(cfg["dims"][1:]):
        raise ValueError("exp.in_channels %s do not match backbone %s" % (inch, name))
    if not getattr(exp, "use_attention", True):
        cfg["n_layer_att"] = 0
    else:
        cfg["n_layer_att"] = int(getattr(exp, "n_layer_att", 3))
    cfg["num_classes"] = int(getattr(exp, "num_classes", 8))
    cfg["embed_dim"] = int(getattr(exp, "embed_dim", 128))
    mask = hasattr(exp, "d_rate") or hasattr(exp, "use_raft")            # ExpTrackMask attributes (unicorn_track_mask.py:44-45)
    cfg["mask"] = bool(mask)
    if mask:
        if not getattr(exp, "use_raft", True) or getattr(exp, "ctrl_loc", "reg") != "reg":
            raise ValueError("unicorn_amd implements the released mask configuration (use_raft=True, ctrl_loc='reg')")
        cfg["d_rate"] = int(getattr(exp, "d_rate", 2))
    return cfg


class HipModelMixin:
    """get_model() of exp/unicorn_track.py:115 / unicorn_track_mask.py:49 returning the HIP drop-in model."""

    hip_precision = "f16x2"

    def get_model(self, load_pretrain=True):
        from ..models import Unicorn
        if getattr(self, "model", None) is None:
            self.model = Unicorn(model_cfg_from_exp(self), precision=getattr(self, "hip_precision", "f16x2"))
            self.model.cfg_name = getattr(self, "exp_name", "custom")
        # load_pretrain (the COCO detector checkpoint, unicorn_track.py:157-190) is a training-time step: checkpoints go through
        # model.load_state_dict / unicorn_amd.utils.checkpoint.load_ckpt like in tools/track.py:186-188
        return self.model


def patch_exp(exp, precision="f16x2"):
    """make an existing (reference) exp object build the HIP model: same object, get_model() swapped"""
    cls = exp.__class__
    exp.__class__ = type(cls.__name__, (HipModelMixin, cls), {"hip_precision": precision})
    return exp


class Exp(HipModelMixin):
    """stand-alone stand-in for exps/default/<name>.py (inference attributes only)"""

    def __init__(self, exp_name, precision="f16x2"):
        backbone, nc, mask, test_size = _DEFAULT_EXPS[exp_name]
        self.exp_name = exp_name
        self.backbone_name = backbone
        self.in_channels = list(_CONVNEXT[backbone]["dims"][1:])
        self.num_classes = nc
        self.embed_dim, self.interact_mode = 128, "deform"
        self.use_attention, self.n_layer_att = True, 3
        self.test_size = self.input_size = test_size
        self.test_conf, self.nmsthre = 0.01, 0.65                     # exp/unicorn_track.py:105-106
        self.normalize = False                                        # :76
        if mask:
            self.use_raft, self.d_rate, self.ctrl_loc = True, 2, "reg"   # unicorn_track_mask.py:38,44-45
        self.hip_precision = precision
        self.model = None


def get_exp(exp_file=None, exp_name=None, precision="f16x2"):
    """unicorn/exp/build.py:35-50.  exp_file wins over exp_name like in the reference."""
    assert exp_file is not None or exp_name is not None, "plz provide exp file or exp name."
    if exp_file is not None:
        name = os.path.basename(exp_file).split(".")[0]
        try:                                    # is the reference tree importable?  (only THIS import may fail quietly)
            importlib.import_module("unicorn.exp")
            have_ref = True
        except ImportError:
            have_ref = False
        if have_ref:                            # use the reference exp file verbatim; an error inside a custom exp file propagates
            sys.path.append(os.path.dirname(exp_file))
            mod = importlib.import_module(name)
            return patch_exp(mod.Exp(), precision)
        exp_name = name
    if exp_name not in _DEFAULT_EXPS:
        raise ValueError("unknown experiment %r (known: %s)" % (exp_name, sorted(_DEFAULT_EXPS)))
    return Exp(exp_name, precision)
