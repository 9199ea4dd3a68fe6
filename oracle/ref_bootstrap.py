"""Bootstrap for importing the REAL reference (`/root/reference`) on CPU.

TEST INFRASTRUCTURE ONLY.  Used by `tests/golden/make_golden.py` (run in the build container,
where /root/reference exists) to generate golden vectors, and by the `not gpu` tests that pin
`oracle/unicorn_oracle.py` against the reference when the reference tree is present.  Nothing on the
product path imports this file; it never runs on the GPU box (the reference is absent there).

Recipe = SURVEY.md Appendix A: stub the third-party imports that are absent offline, redirect
device="cuda" factory calls to CPU, and route MSDeformAttnFunction.apply to the reference's own
pure-PyTorch `ms_deform_attn_core_pytorch` (the reference CUDA op has no CPU implementation,
unicorn/models/ops/src/ms_deform_attn.h:19-38).
"""
import os
import sys
import types

REF_ROOT = os.environ.get("UNICORN_REFERENCE", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "unicorn", "models"))


class _Anything:
    """Object whose every attribute/call is a no-op (loguru.logger stand-in)."""

    def __getattr__(self, name):
        return _Anything()

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]  # decorator usage: @logger.catch
        return _Anything()


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _install_stubs():
    import torch
    import torch.nn as nn

    _mod("loguru", logger=_Anything())

    class DropPath(nn.Module):
        def __init__(self, p=0.0):
            super().__init__()

        def forward(self, x):
            return x

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    timm = _mod("timm")
    timm.models = _mod("timm.models")
    timm.models.layers = _mod("timm.models.layers", trunc_normal_=nn.init.trunc_normal_,
                              DropPath=DropPath, to_2tuple=to_2tuple)

    # torchvision.ops restated (greedy NMS, suppress IoU > thr; batched = coordinate offset trick)
    def box_iou(a, b):
        area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
        area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        lt = torch.max(a[:, None, :2], b[None, :, :2])
        rb = torch.min(a[:, None, 2:], b[None, :, 2:])
        wh = (rb - lt).clamp(min=0)
        inter = wh[..., 0] * wh[..., 1]
        return inter / (area_a[:, None] + area_b[None] - inter)

    def nms(boxes, scores, thr):
        order = torch.argsort(scores, descending=True, stable=True)
        keep = []
        sup = torch.zeros(len(boxes), dtype=torch.bool)
        iou = box_iou(boxes, boxes) if len(boxes) else None
        for i in order.tolist():
            if sup[i]:
                continue
            keep.append(i)
            sup |= iou[i] > thr
        return torch.tensor(keep, dtype=torch.long)

    def batched_nms(boxes, scores, idxs, thr):
        if boxes.numel() == 0:
            return torch.empty((0,), dtype=torch.long)
        off = idxs.to(boxes) * (boxes.max() + 1)
        return nms(boxes + off[:, None], scores, thr)

    tv = _mod("torchvision")
    tv.ops = _mod("torchvision.ops", nms=nms, batched_nms=batched_nms, box_iou=box_iou)

    # ByteTrack's third-party solvers (absent offline): restated from their published semantics in oracle/bytetrack_oracle.py
    def _lapjv(cost, extend_cost=True, cost_limit=float("inf")):
        import bytetrack_oracle as _bo
        return _bo.lapjv(cost, extend_cost, cost_limit)

    def _bbox_overlaps(a, b):
        import bytetrack_oracle as _bo
        return _bo.bbox_overlaps(a, b)

    _mod("lap", lapjv=_lapjv)
    _mod("cython_bbox", bbox_overlaps=_bbox_overlaps)
    import numpy as _np
    for _alias, _t in (("float", float), ("int", int), ("bool", bool)):     # removed in numpy >= 1.24, used by the reference
        if not hasattr(_np, _alias):
            setattr(_np, _alias, _t)

    _mod("thop", profile=lambda *a, **k: (0, 0))
    cv2 = _mod("cv2", setNumThreads=lambda n: None)
    cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda b: None)

    class AttrDict(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

        def clone(self):
            import copy
            return copy.deepcopy(self)

    class CfgNode(AttrDict):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)

    _mod("yacs", config=_mod("yacs.config", CfgNode=CfgNode))

    class EasyDict(AttrDict):
        def __init__(self, d=None, **kw):
            super().__init__()
            d = dict(d or {}, **kw)
            for k, v in d.items():
                self[k] = EasyDict(v) if isinstance(v, dict) and not isinstance(v, EasyDict) else v

        def __setattr__(self, k, v):
            self[k] = EasyDict(v) if isinstance(v, dict) and not isinstance(v, EasyDict) else v

    _mod("easydict", EasyDict=EasyDict)
    _mod("MultiScaleDeformableAttention")
    pc = _mod("pycocotools")
    pc.mask = _mod("pycocotools.mask")
    pc.coco = _mod("pycocotools.coco", COCO=object)
    pc.cocoeval = _mod("pycocotools.cocoeval", COCOeval=object)
    _mod("motmetrics")
    _mod("tabulate", tabulate=lambda *a, **k: "") if "tabulate" not in sys.modules else None


def _redirect_cuda_factories():
    import torch
    if torch.cuda.is_available():
        return
    for name in ("zeros", "ones", "full", "tensor", "arange", "empty"):
        orig = getattr(torch, name)
        if getattr(orig, "_uni_wrapped", False):
            continue

        def make(o):
            def wrapped(*a, **k):
                if str(k.get("device", "")).startswith("cuda"):
                    k["device"] = "cpu"
                return o(*a, **k)
            wrapped._uni_wrapped = True
            return wrapped
        setattr(torch, name, make(orig))


_BOOTED = False


def boot():
    """Make `import unicorn` resolve to the reference. Idempotent."""
    global _BOOTED
    if _BOOTED:
        return
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    _install_stubs()
    import torch  # noqa
    _redirect_cuda_factories()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import unicorn.models  # noqa
    import unicorn.models.ops.modules.ms_deform_attn as mm
    from unicorn.models.ops.functions.ms_deform_attn_func import ms_deform_attn_core_pytorch as core
    mm.MSDeformAttnFunction = type(
        "F", (), {"apply": staticmethod(lambda v, shp, lsi, loc, w, step: core(v, shp, loc, w))})
    _BOOTED = True


def build_reference_model(exp_name):
    """exp_name e.g. 'unicorn_track_tiny'. Returns the reference nn.Module in eval mode (CPU)."""
    boot()
    import warnings
    from unicorn.exp import get_exp
    cwd = os.getcwd()
    os.chdir(REF_ROOT)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            exp = get_exp("exps/default/%s.py" % exp_name, None)
            model = exp.get_model(load_pretrain=False)
    finally:
        os.chdir(cwd)
    model.eval()
    # stages are checkpoint-wrapped even in eval (exp/unicorn_track.py:125-126); harmless but noisy
    model.backbone.backbone.use_checkpoint = False
    return model, exp
