"""`unicorn.tracker.quasi_dense_embed_tracker.QuasiDenseEmbedTracker` with the reference's constructor and `match`
signature (unicorn/tracker/quasi_dense_embed_tracker.py:9-42,137-212), backed by the native library behind
include/unicorn_assoc.h (unicorn_amd/csrc/assoc.cpp).  No python fallback: a missing library raises.
"""
import ctypes as C
import os

import numpy as np
import torch

_LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lib", "libunicorn_assoc.so")
_METRICS = {"bisoftmax": 0, "softmax": 1, "cosine": 2}


class UnicornAssocError(RuntimeError):
    pass


class _Cfg(C.Structure):
    _fields_ = [("init_score_thr", C.c_float), ("obj_score_thr", C.c_float), ("match_score_thr", C.c_float),
                ("memo_tracklet_frames", C.c_int32), ("memo_backdrop_frames", C.c_int32), ("memo_momentum", C.c_float),
                ("nms_conf_thr", C.c_float), ("nms_backdrop_iou_thr", C.c_float), ("nms_class_iou_thr", C.c_float),
                ("with_cats", C.c_int32), ("match_metric", C.c_int32)]


_lib = None


def assoc_lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise UnicornAssocError("native association library missing: %s (run __graft_entry__.build())" % _LIB_PATH)
        from .._lib import verify_manifest, UnicornHipError
        try:
            verify_manifest("libunicorn_assoc.so")      # built from this tree's assoc.cpp (csrc/build.sh manifest)
        except UnicornHipError as e:
            raise UnicornAssocError(str(e))
        L = C.CDLL(_LIB_PATH)
        L.uni_qd_default_cfg.argtypes = [C.POINTER(_Cfg)]
        L.uni_qd_default_cfg.restype = None
        L.uni_qd_create.argtypes = [C.POINTER(_Cfg)]
        L.uni_qd_create.restype = C.c_void_p
        L.uni_qd_destroy.argtypes = [C.c_void_p]
        L.uni_qd_destroy.restype = None
        L.uni_qd_last_error.restype = C.c_char_p
        L.uni_qd_match.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.uni_qd_match.restype = C.c_int
        L.uni_qd_num_tracklets.argtypes = [C.c_void_p]
        L.uni_qd_num_tracklets.restype = C.c_int64
        L.uni_qd_alive.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.uni_qd_alive.restype = C.c_int
        _lib = L
    return _lib


class QuasiDenseEmbedTracker(object):
    def __init__(self, init_score_thr=0.8, obj_score_thr=0.5, match_score_thr=0.5, memo_tracklet_frames=30,
                 memo_backdrop_frames=1, memo_momentum=0.8, nms_conf_thr=0.5, nms_backdrop_iou_thr=0.3,
                 nms_class_iou_thr=0.7, with_cats=True, match_metric="bisoftmax"):
        assert 0 <= memo_momentum <= 1.0
        assert memo_tracklet_frames >= 0
        assert memo_backdrop_frames >= 0
        assert match_metric in ["bisoftmax", "softmax", "cosine"]
        cfg = _Cfg(init_score_thr, obj_score_thr, match_score_thr, memo_tracklet_frames, memo_backdrop_frames, memo_momentum,
                   nms_conf_thr, nms_backdrop_iou_thr, nms_class_iou_thr, int(bool(with_cats)), _METRICS[match_metric])
        self._h = assoc_lib().uni_qd_create(C.byref(cfg))
        if not self._h:
            raise UnicornAssocError(assoc_lib().uni_qd_last_error().decode())

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.uni_qd_destroy(h)

    @property
    def num_tracklets(self):
        return int(assoc_lib().uni_qd_num_tracklets(self._h))

    @property
    def tracklet_ids(self):
        buf = np.empty(max(self.num_tracklets, 1), dtype=np.int64)
        n = assoc_lib().uni_qd_alive(self._h, buf.ctypes.data, buf.size)
        return buf[:n].tolist()

    @property
    def empty(self):
        buf = np.empty(1, dtype=np.int64)
        return assoc_lib().uni_qd_alive(self._h, buf.ctypes.data, 0) == 0

    def match(self, bboxes, labels, track_feats, frame_id, asso_tau=-1, return_index=False):
        """bboxes (N,5) [x1,y1,x2,y2,score], labels (N,), track_feats (N,D) -> (bboxes, labels, ids[, valids]) on the CPU,
        exactly like the reference (its callers pass .cpu() tensors, mot_evaluator.py:1041-1045)."""
        b = np.ascontiguousarray(bboxes.detach().cpu().numpy(), dtype=np.float32)
        lab = np.ascontiguousarray(labels.detach().cpu().numpy(), dtype=np.int64)
        f = np.ascontiguousarray(track_feats.detach().cpu().numpy(), dtype=np.float32)
        n = b.shape[0]
        if b.ndim != 2 or (n and b.shape[1] != 5) or lab.shape[0] != n or f.shape[0] != n:
            raise ValueError("match: bboxes (N,5), labels (N,), track_feats (N,D) expected")
        dim = f.shape[1] if f.ndim == 2 else 0
        ob, ol, oi = np.empty((max(n, 1), 5), np.float32), np.empty(max(n, 1), np.int64), np.empty(max(n, 1), np.int64)
        ov = np.zeros(max(n, 1), np.uint8)
        m = C.c_int(0)
        rc = assoc_lib().uni_qd_match(self._h, b.ctypes.data, lab.ctypes.data, f.ctypes.data, n, dim, int(frame_id), ob.ctypes.data,
                                      ol.ctypes.data, oi.ctypes.data, ov.ctypes.data, C.byref(m))
        if rc != 0:
            raise UnicornAssocError(assoc_lib().uni_qd_last_error().decode())
        k = m.value
        out = (torch.from_numpy(ob[:k].copy()), torch.from_numpy(ol[:k].copy()), torch.from_numpy(oi[:k].copy()))
        if return_index:
            return out + (torch.from_numpy(ov[:n].astype(np.bool_)),)
        return out
