"""Drop-in for ``unicorn.models.Unicorn`` (inference modes only) running on libunicorn_hip.so.

API mirrored from the reference (paths relative to MasterBin-IIAU/Unicorn):
  Unicorn.forward(mode="backbone"|"interaction"|"upsample"|"whole")   unicorn/models/unicorn.py:60-74,133-139
  model.head(fpn_outs, priors, mode="sot"|"mot")                       unicorn_head.py:249 / unicorn_head_mask.py:280
  model.head.mask_head(mask_feats, 8, mask_head_params=..., ...)       condinst/dynamic_mask_head.py:227-285
  model.load_state_dict(sd, strict=False) -> (missing_keys, unexpected_keys), .eval(), .cuda(), .half()
Tensors returned to the caller are ordinary torch tensors of the reference's NCHW *shape*; their memory
is channels_last (NHWC), which is what the HIP kernels produce/consume, so round trips are copy-free.
"""
import ctypes as C
import os
from collections import namedtuple

import numpy as np
import torch

from .. import _lib as L
from ..ops import condinst_masks, empty_nhwc, nhwc

PRECISIONS = {"bf16": 0, "fp32": 1, "f16x2": 2}      # uni_model_cfg.precision
_IncompatibleKeys = namedtuple("IncompatibleKeys", ["missing_keys", "unexpected_keys"])

# exps/default/*.py + unicorn/exp/unicorn_track.py:31-113, unicorn_track_mask.py:31-47
MODEL_CONFIGS = {
    "unicorn_track_tiny": dict(dims=(96, 192, 384, 768), depths=(3, 3, 9, 3), num_classes=8, mask=False),
    "unicorn_track_tiny_mask": dict(dims=(96, 192, 384, 768), depths=(3, 3, 9, 3), num_classes=8, mask=True),
    "unicorn_track_large": dict(dims=(192, 384, 768, 1536), depths=(3, 3, 27, 3), num_classes=8, mask=False),
    "unicorn_track_large_mask": dict(dims=(192, 384, 768, 1536), depths=(3, 3, 27, 3), num_classes=8, mask=True),
    "unicorn_track_large_mot_challenge": dict(dims=(192, 384, 768, 1536), depths=(3, 3, 27, 3), num_classes=1, mask=False),
    "unicorn_track_large_mot_challenge_mask": dict(dims=(192, 384, 768, 1536), depths=(3, 3, 27, 3), num_classes=1, mask=True),
}


class DynamicMaskHead(torch.nn.Module):
    """condinst/dynamic_mask_head.py: callable returning sigmoid mask scores (N,1,up_rate*H8,up_rate*W8)."""

    def __init__(self, up_rate):
        super().__init__()
        self.up_rate = up_rate
        self.training = False

    def forward(self, mask_feats, mask_feat_stride, mask_head_params=None, instance_locations=None,
                 instance_fpn_levels=None, gt_bitmasks=None, up_masks=None):
        if mask_feat_stride != 8 or up_masks is None:
            raise ValueError("DynamicMaskHead: only mask_feat_stride=8 with RAFT up_masks is supported (use_raft=True)")
        return condinst_masks(mask_feats, up_masks, mask_head_params, instance_locations, instance_fpn_levels,
                              self.up_rate, 1)


class UnicornHead(torch.nn.Module):
    """unicorn_head.py:19-482 (eval branch). Owned by Unicorn; shares its context."""

    def __init__(self, model):
        super().__init__()
        object.__setattr__(self, "_m", model)          # back reference, NOT a registered submodule (the model owns the head)
        self.num_classes = model.num_classes
        self.num_classes_sot = 1
        self.decode_in_inference = True
        self.strides = [8, 16, 32]
        self.training = False
        self.hw = None

    def _run(self, xin, mask_in, mode):
        if mode not in ("sot", "mot"):
            raise ValueError("""mode has to be 'sot' or 'mot'""")        # unicorn_head.py:292
        raw = not self.decode_in_inference          # tools/export_torchscript.py:66: undecoded rows (unicorn_head.py:436-439)
        if raw and self._m.mask:
            raise ValueError()                       # unicorn_head_mask.py:470: the mask head has no undecoded path
        m = self._m
        m._require_ready()
        f = [nhwc(x) for x in xin]
        Bp = mask_in[0].shape[0]
        Bi = f[0].shape[0]
        objects = Bi == 1 and Bp > 1           # object-batched call (row N3): ONE image, Bp prior sets (uni_head_objects)
        B = Bp if objects else Bi
        H, W = f[0].shape[2] * 8, f[0].shape[3] * 8
        pri = []
        for k, p in enumerate(mask_in):
            if p.shape[1] != 1 or p.shape[0] != B:
                raise ValueError("head: priors must be (B,1,H/s,W/s) (or (K,1,H/s,W/s) prior sets for ONE image)")
            pri.append(p.float().contiguous())
        nc = 1 if mode == "sot" else self.num_classes
        A = sum(x.shape[2] * x.shape[3] for x in f)
        dev = f[0].device
        out = torch.empty((B, A, 5 + nc), device=dev, dtype=torch.float32)
        dyn = mf = um = None
        if m.mask:
            dyn = torch.empty((B, A, 169), device=dev, dtype=torch.float32)
            mf = empty_nhwc(8, H // 8, W // 8, dev, Bi)                 # mask branch outputs depend on the image only
            um = empty_nhwc(9 * m.up_rate ** 2, H // 8, W // 8, dev, Bi)
        fn = L.lib().uni_head_objects if objects else L.lib().uni_head
        L.check(fn(m._ctx, L.ptr(f[0]), L.ptr(f[1]), L.ptr(f[2]), L.ptr(pri[0]), L.ptr(pri[1]), L.ptr(pri[2]),
                   B, H, W, (0 if mode == "sot" else 1) | (2 if raw else 0), L.ptr(out), L.ptr(dyn), L.ptr(mf), L.ptr(um),
                   L.stream_ptr()), "uni_head_objects" if objects else "uni_head")
        self.hw = [tuple(x.shape[2:]) for x in f]
        return out, dyn, mf, um

    def forward(self, xin, mask_in, labels=None, imgs=None, mode=None, **kw):
        return self._run(xin, mask_in, mode)[0]

    def decode_outputs(self, outputs, dtype=None):
        """unicorn_head.py:467-482 on raw (B,A,5+nc) outputs (in place, uni_decode_outputs)."""
        if not (outputs.is_cuda and outputs.dtype == torch.float32 and outputs.is_contiguous()):
            raise ValueError("decode_outputs: needs a contiguous float32 cuda tensor (B, A, 5+nc)")
        H, W = self.hw[0][0] * 8, self.hw[0][1] * 8
        L.check(L.lib().uni_decode_outputs(L.ptr(outputs), outputs.shape[0], H, W, outputs.shape[2], L.stream_ptr()), "uni_decode_outputs")
        return outputs


class UnicornHeadMask(UnicornHead):
    """unicorn_head_mask.py:22-519 (eval branch) -> (outputs, locations, dynamic_params, fpn_levels, mask_feats, up_masks)."""

    def __init__(self, model):
        super().__init__(model)
        self.mask_head = DynamicMaskHead(model.up_rate)
        self._levels_cache = {}

    def forward(self, xin, mask_in, labels=None, imgs=None, mode=None, **kw):
        out, dyn, mf, um = self._run(xin, mask_in, mode)
        grids, strides = self._m._grids(self.hw, out.device)
        locations = ((grids + 0.5) * strides)[0]                               # unicorn_head_mask.py:518
        # fpn_levels live on the CPU like the reference's (unicorn_head_mask.py:519); they depend on the shapes only, so they are
        # built once: a (K, 21000) torch.full / cat per call is large enough to wake torch's OpenMP team, whose spin-waiting
        # threads can exhaust a container's CPU quota and freeze the HIP dispatch thread for tens of ms (seen as periodic stalls)
        key = (out.shape[0], tuple(self.hw))
        if key not in self._levels_cache:
            self._levels_cache[key] = torch.cat([torch.full((out.shape[0], h * w), k) for k, (h, w) in enumerate(self.hw)], 1)
        return out, locations, dyn, self._levels_cache[key], mf, um


class Unicorn(torch.nn.Module):
    """A `torch.nn.Module` SHELL around the HIP context: the reference's tools treat the model as a Module (`DDP(model, ...)`
    tools/track.py:193-194, `model.module.head` external/qdtrack/qdtrack/apis/test_omni.py:77,90, `.eval()/.cuda()/.half()`), so the
    class is one -- `.head` is a registered submodule, `.eval() / .train(False) / .modules() / next(model.parameters()).device` work,
    and one 1-element `_hip_anchor` parameter (untouched by the forward) lets `DistributedDataParallel` wrap it.  The weights
    themselves live re-packed inside the context (csrc/engine.hip), not as torch parameters: `state_dict()` returns the anchor only,
    `load_state_dict` takes the reference checkpoint namespace.  `torch.jit.trace` cannot see through the C-ABI calls and raises."""

    _half_warned = False      # .half() explains itself once per process

    def __init__(self, name_or_cfg="unicorn_track_tiny", device=None, precision="f16x2"):
        """precision (operand format of every dense contraction; accumulation, residual stream and statistics are fp32):
          "f16x2"  (default) fp32-equivalent: operands split into hi + lo f16 halves, 3 f16 MFMAs per product (22 operand bits).
                   Meets box/mask IoU >= 0.999 and embedding cosine within 1e-4 against the fp32 reference: parity mode.
                   Operand range: values entering a contraction saturate at +-65504 (the residual stream, statistics and
                   everything crossing the API stay fp32); `check_saturation(True)` + `saturation_stats()` count saturated
                   operands when validating a new checkpoint.
          "fp32"   exact fp32 MFMA (v_mfma_f32_32x32x2_f32), bitwise an fmaf chain; 16x the bf16 MFMA cost.
          "bf16"   bf16 operands, 1 MFMA per product; fastest, embedding cosine still within 1e-4 but box IoU is NOT
                   (8 operand bits; profiles/r02_precision_budget.json)."""
        super().__init__()
        if precision not in PRECISIONS:
            raise ValueError("precision must be one of %s" % (sorted(PRECISIONS),))
        self.precision = precision
        self._hip_anchor = torch.nn.Parameter(torch.zeros(1))
        cfg = dict(MODEL_CONFIGS[name_or_cfg]) if isinstance(name_or_cfg, str) else dict(name_or_cfg)
        self.cfg_name = name_or_cfg if isinstance(name_or_cfg, str) else "custom"
        self.dims, self.depths = tuple(cfg["dims"]), tuple(cfg["depths"])
        self.num_classes, self.mask = int(cfg["num_classes"]), bool(cfg["mask"])
        self.n_layer_att = int(cfg.get("n_layer_att", 3))
        self.embed_dim = int(cfg.get("embed_dim", 128))
        self.d_rate = int(cfg.get("d_rate", 2))
        self.up_rate = 8 // self.d_rate
        self.interact_mode = "deform"
        self.training = False
        self._ctx = None
        self._ready = False
        self._device = None
        self._pos_cache = {}
        self._grid_cache = {}
        self.head = UnicornHeadMask(self) if self.mask else UnicornHead(self)
        if device is not None:
            self.cuda(device)

    # ---------------------------------------------------------------- nn.Module surface (overrides: the weights are not torch tensors)
    def cuda(self, device=None):
        if not torch.cuda.is_available():
            raise L.UnicornHipError("unicorn_amd needs a HIP device (torch.cuda.is_available() is False); no CPU fallback")
        idx = torch.cuda.current_device() if device is None else torch.device("cuda", device).index if isinstance(device, int) \
            else torch.device(device).index
        idx = 0 if idx is None else idx
        if self._ctx is not None and self._device is not None and self._device.index != idx:
            raise L.UnicornHipError("model already bound to %s" % self._device)
        self._device = torch.device("cuda", idx)
        self._hip_anchor.data = self._hip_anchor.data.to(self._device)
        if self._ctx is None:
            c = L.ModelCfg()
            c.dims[:] = self.dims
            c.depths[:] = self.depths
            c.num_classes, c.mask, c.n_layer_att = self.num_classes, int(self.mask), self.n_layer_att
            c.embed_dim, c.up_rate, c.d_rate = self.embed_dim, self.up_rate, self.d_rate
            c.precision = PRECISIONS[self.precision]
            ctx = L.lib().uni_ctx_create(idx, C.byref(c))
            if not ctx:
                raise L.UnicornHipError("uni_ctx_create: %s" % L.lib().uni_last_error().decode())
            self._ctx = C.c_void_p(ctx)
            if getattr(self, "_pending_sd", None) is not None:
                sd, self._pending_sd = self._pending_sd, None
                self._load(sd)
        return self

    def to(self, device=None, *a, **k):
        if device is None or isinstance(device, torch.dtype):      # .to(dtype): the operand format is fixed by `precision`
            return self
        return self.cuda(device)

    def train(self, mode=True):
        if mode:
            raise L.UnicornHipError("unicorn_amd implements the inference path only")
        return super().train(False)                                 # .eval() == .train(False)

    def state_dict(self, *a, **k):
        return super().state_dict(*a, **k)                          # the anchor only; the packed weights are device-side and immutable

    def half(self):
        # tools/track.py --fp16 calls model.half() and feeds half images (mot_evaluator.py:126-128): the operand format of the HIP
        # path is fixed by `precision` at construction, inputs are widened to fp32 (forward_backbone), so this is a no-op -- said ONCE,
        # because a --fp16 user otherwise pays the fp32-equivalent cost without knowing why nothing got faster
        if not Unicorn._half_warned:
            Unicorn._half_warned = True
            import warnings
            warnings.warn("unicorn_amd: model.half() does not change the arithmetic -- the operand format is fixed by `precision=%r` at "
                          "construction (f16x2 = fp32-equivalent split-f16 MFMA operands; half inputs are widened to fp32).  A single-f16 or "
                          "bf16 path cannot meet the box IoU >= 0.999 parity bar (DESIGN.md section 2)." % (self.precision,), stacklevel=2)
        return self

    def check_saturation(self, on=True):
        """uni_ctx_set_check: count f16x2 operands that hit the +-65504 saturation bound (resets the counters).  In check mode the
        ConvNeXt MLPs run as the two-launch pair (pwconv1+GELU, pwconv2) so that the hidden operand tensor exists to be scanned: the
        fused-MLP kernel of the production path forms the same hidden operands in registers (same split, same saturation), its
        outputs agree with the pair to fp32 round-off but are not bit-identical (tests: test_mlp_fused_matches_unfused_pair)."""
        self._require_ready()
        L.check(L.lib().uni_ctx_set_check(self._ctx, int(bool(on))), "uni_ctx_set_check")
        return self

    def saturation_stats(self):
        self._require_ready()
        buf = (C.c_longlong * 4)()
        L.check(L.lib().uni_ctx_stats(self._ctx, buf), "uni_ctx_stats")
        return {"saturated": int(buf[0]), "scanned": int(buf[1]), "buffers": int(buf[2])}

    def float(self):
        return self

    def __del__(self):
        try:
            if self._ctx is not None:
                L.lib().uni_ctx_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass

    def load_state_dict(self, state_dict, strict=True):
        """Same contract as nn.Module.load_state_dict; keys follow the reference checkpoint namespace."""
        if self._ready:
            raise L.UnicornHipError("weights already loaded (the packed device copy is immutable)")
        res = self._check_keys(state_dict, strict)          # spec-based, on the host: same answer before and after .cuda()
        if self._ctx is None:                               # reference order is load -> cuda (tools/track.py:176-188): upload later
            self._pending_sd = state_dict
            return res
        self._load(state_dict)
        return res

    def load_flat_file(self, path):
        """weights from a flat file written by `unicorn_amd.utils.checkpoint.export_flat` (uni_ctx_load_file: the path a host without torch takes); the file's
        network configuration must be this model's"""
        if self._ready:
            raise L.UnicornHipError("weights already loaded (the packed device copy is immutable)")
        if self._ctx is None:
            raise L.UnicornHipError("call model.cuda() first (unicorn_amd has no CPU path)")
        lib = L.lib()
        n = C.c_int(0)
        L.check(lib.uni_ctx_load_file(self._ctx, os.fsencode(path), C.byref(n)), "uni_ctx_load_file")
        nm = C.c_int(0)
        L.check(lib.uni_ctx_finalize(self._ctx, C.byref(nm)), "uni_ctx_finalize")
        self._engine_missing = [lib.uni_ctx_missing_name(self._ctx, i).decode() for i in range(nm.value)]
        self._ready = True
        return n.value

    def _check_keys(self, state_dict, strict):
        """missing / unexpected keys against the reference model's parameter spec; a tensor whose SHAPE differs raises like
        nn.Module.load_state_dict does (a wrong-config checkpoint must not run on zero weights)."""
        from ..utils.checkpoint import state_spec      # names the reference model owns (buffers / foreign keys are "unexpected")
        spec = state_spec(dict(dims=self.dims, depths=self.depths, num_classes=self.num_classes, mask=self.mask,
                               n_layer_att=self.n_layer_att, embed_dim=self.embed_dim, up_rate=self.up_rate))
        fl = {k: v for k, v in state_dict.items() if torch.is_tensor(v) and v.dtype.is_floating_point}
        bad = ["%s: checkpoint %s vs model %s" % (k, tuple(v.shape), tuple(spec[k])) for k, v in fl.items()
               if k in spec and tuple(v.shape) != tuple(spec[k])]
        if bad:
            raise RuntimeError("Error(s) in loading state_dict: size mismatch for " + "; ".join(bad[:8]))
        missing = [k for k in spec if k not in fl]
        unexpected = [k for k in fl if k not in spec]
        if strict and (missing or unexpected):
            raise RuntimeError("Error(s) in loading state_dict: missing keys %s, unexpected keys %s" % (missing[:8], unexpected[:8]))
        return _IncompatibleKeys(missing, unexpected)

    def _load(self, state_dict):
        lib = L.lib()
        for k, v in state_dict.items():
            if not torch.is_tensor(v) or not v.dtype.is_floating_point:
                continue
            a = np.ascontiguousarray(v.detach().float().cpu().numpy())
            shp = (C.c_int64 * max(a.ndim, 1))(*a.shape)
            L.check(lib.uni_ctx_load_param(self._ctx, k.encode(), a.ctypes.data_as(C.c_void_p), shp, a.ndim), "load_param")
        nm = C.c_int(0)
        L.check(lib.uni_ctx_finalize(self._ctx, C.byref(nm)), "uni_ctx_finalize")
        self._engine_missing = [lib.uni_ctx_missing_name(self._ctx, i).decode() for i in range(nm.value)]   # zero-filled by the engine
        self._ready = True

    def _require_ready(self):
        if self._ctx is None:
            raise L.UnicornHipError("call model.cuda() first (unicorn_amd has no CPU path)")
        if not self._ready:
            raise L.UnicornHipError("load_state_dict() must be called before running the model")

    # ---------------------------------------------------------------- helpers
    def _grids(self, hw, device):
        key = (tuple(hw), str(device))
        if key not in self._grid_cache:
            grids, strides = [], []
            for (h, w), s in zip(hw, (8, 16, 32)):
                yv, xv = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
                grids.append(torch.stack((xv, yv), 2).view(1, -1, 2).float())
                strides.append(torch.full((1, h * w, 1), float(s)))
            self._grid_cache[key] = (torch.cat(grids, 1).to(device), torch.cat(strides, 1).to(device))
        return self._grid_cache[key]

    def _pos(self, h, w):
        key = (h, w)
        if key not in self._pos_cache:      # constant per resolution
            p = empty_nhwc(256, h, w, self._device)
            L.check(L.lib().uni_pos_embed(self._ctx, h, w, L.ptr(p), L.stream_ptr()), "uni_pos_embed")
            self._pos_cache[key] = p
        return self._pos_cache[key]

    # ---------------------------------------------------------------- forward modes
    def forward(self, imgs=None, run_fpn=True, seq_dict0=None, seq_dict1=None, feat=None, mode="whole", **unused):
        if torch.jit.is_tracing():
            raise L.UnicornHipError("torch.jit.trace cannot record the C-ABI calls of unicorn_amd (tools/export_torchscript.py:70); "
                                    "use head.decode_in_inference = False for the raw export rows instead")
        if mode == "backbone":
            return self.forward_backbone(imgs, run_fpn)
        elif mode == "interaction":
            return self.forward_deform_interact(seq_dict0, seq_dict1)
        elif mode == "upsample":
            return self.forward_upsample(feat)
        elif mode == "whole":                                              # unicorn.py:133-139
            bs, _, H, W = imgs.size()
            fpn_outs, seq_dict = self.forward_backbone(imgs, run_fpn=True)
            dev = fpn_outs[0].device
            pri = tuple(torch.zeros((bs, 1, H // s, W // s), device=dev) for s in (8, 16, 32))
            return self.head(fpn_outs, pri, mode="mot"), seq_dict
        else:
            raise ValueError                                               # unicorn.py:229

    def forward_backbone(self, img, run_fpn=True):
        assert isinstance(img, torch.Tensor)
        self._require_ready()
        if img.dim() != 4 or img.shape[1] != 3:
            raise ValueError("forward_backbone expects a (B,3,H,W) image batch")
        if not img.is_cuda:
            raise L.UnicornHipError("input image must be a HIP device tensor; no CPU fallback")
        x = img.float().contiguous()
        B, _, H, W = x.shape
        dev = x.device
        c1, c2, c3 = self.dims[1:]
        fpn = (empty_nhwc(c1, H // 8, W // 8, dev, B), empty_nhwc(c2, H // 16, W // 16, dev, B), empty_nhwc(c3, H // 32, W // 32, dev, B))
        feat16 = empty_nhwc(c2, H // 16, W // 16, dev, B)
        L.check(L.lib().uni_backbone_fpn(self._ctx, L.ptr(x), B, H, W, L.ptr(fpn[0]), L.ptr(fpn[1]), L.ptr(fpn[2]), L.ptr(feat16),
                                         L.stream_ptr()), "uni_backbone_fpn")
        h, w = H // 16, W // 16
        seq_dict = {"feat": feat16, "pos": self._pos(h, w).expand(B, -1, -1, -1), "h": h, "w": w}
        return (fpn, seq_dict) if run_fpn else seq_dict

    def forward_deform_interact(self, d0, d1):
        self._require_ready()
        f0, f1 = nhwc(d0["feat"]), nhwc(d1["feat"])
        p0, p1 = nhwc(d0["pos"][0:1]), nhwc(d1["pos"][0:1])      # the learned position embedding is batch-invariant
        h, w = d0["h"], d0["w"]
        B = f1.shape[0]
        if f0.shape[0] == 1 and B > 1:                             # one reference frame for a batch of current frames
            f0 = nhwc(f0.expand(B, -1, -1, -1))
        if tuple(f1.shape) != tuple(f0.shape):
            raise ValueError("interaction: reference and current feature maps must have the same shape")
        o0, o1 = empty_nhwc(256, h, w, f0.device, B), empty_nhwc(256, h, w, f0.device, B)
        L.check(L.lib().uni_interaction(self._ctx, L.ptr(f0), L.ptr(p0), L.ptr(f1), L.ptr(p1), B, h, w, L.ptr(o0), L.ptr(o1),
                                        L.stream_ptr()), "uni_interaction")
        return o0, o1

    def forward_upsample(self, x):
        self._require_ready()
        f = nhwc(x)
        B, c, h, w = f.shape
        if c != 256:
            raise ValueError("upsample expects a 256-channel map")
        e = empty_nhwc(self.embed_dim, 2 * h, 2 * w, f.device, B)
        L.check(L.lib().uni_upsample(self._ctx, L.ptr(f), B, h, w, L.ptr(e), L.stream_ptr()), "uni_upsample")
        return e
