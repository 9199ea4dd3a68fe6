"""`unicorn.tracker.byte_tracker.BYTETracker` with the reference's constructor and `update` signature
(unicorn/tracker/byte_tracker.py:142-293), backed by the native library behind include/unicorn_assoc.h
(unicorn_amd/csrc/assoc.cpp: Kalman filter, IoU cost, score fusion, exact linear assignment, track management).
No python fallback: a missing library raises.
"""
import ctypes as C

import numpy as np

from .quasi_dense_embed_tracker import UnicornAssocError, assoc_lib


class _Cfg(C.Structure):
    _fields_ = [("track_thresh", C.c_float), ("track_buffer", C.c_int32), ("match_thresh", C.c_float), ("mot20", C.c_int32),
                ("frame_rate", C.c_int32)]


_bound = False


def _lib():
    global _bound
    L = assoc_lib()
    if not _bound:
        L.uni_byte_create.argtypes = [C.POINTER(_Cfg)]
        L.uni_byte_create.restype = C.c_void_p
        L.uni_byte_destroy.argtypes = [C.c_void_p]
        L.uni_byte_destroy.restype = None
        L.uni_byte_update.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.uni_byte_update.restype = C.c_int
        L.uni_byte_id_count.restype = C.c_int64
        L.uni_byte_clean_id.restype = None
        L.uni_byte_lost.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.uni_byte_lost.restype = C.c_int
        _bound = True
    return L


class STrack(object):
    """what the drivers read from the tracks `update` returns (mot_evaluator.py:213-222): tlwh, track_id, score"""
    __slots__ = ("tlwh", "track_id", "score")

    def __init__(self, tlwh, track_id, score):
        self.tlwh, self.track_id, self.score = tlwh, track_id, score

    @property
    def tlbr(self):
        r = self.tlwh.copy()
        r[2:] += r[:2]
        return r

    def __repr__(self):
        return "OT_{}".format(self.track_id)


def clean_id():
    """BaseTrack.clean_id() (basetrack.py:58-60): the id counter is process-wide in the reference"""
    _lib().uni_byte_clean_id()


def id_count():
    return int(_lib().uni_byte_id_count())


class BYTETracker(object):
    def __init__(self, args, frame_rate=30):
        self.args = args
        cfg = _Cfg(float(args.track_thresh), int(args.track_buffer), float(args.match_thresh), int(bool(args.mot20)), int(frame_rate))
        self._h = _lib().uni_byte_create(C.byref(cfg))
        if not self._h:
            raise UnicornAssocError(assoc_lib().uni_qd_last_error().decode())
        self.frame_id = 0

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                assoc_lib().uni_byte_destroy(h)
            except Exception:
                pass

    @property
    def lost_ids(self):
        buf = np.empty(4096, dtype=np.int64)
        n = _lib().uni_byte_lost(self._h, buf.ctypes.data, buf.size)
        return buf[:min(n, buf.size)].tolist()

    def update(self, output_results, img_info, img_size):
        """output_results: (N,5) ndarray [x1,y1,x2,y2,score] or (N,>=6) tensor / ndarray (score = c4*c5), network-input
        coordinates; img_info = (img_h, img_w, ...); img_size = (H, W) of the network input.  Returns the activated tracks."""
        if hasattr(output_results, "detach"):
            output_results = output_results.detach().cpu().numpy()
        d = np.ascontiguousarray(output_results, dtype=np.float32)
        if d.ndim != 2 or (d.shape[0] and d.shape[1] < 5):
            raise ValueError("update: output_results must be (N,5) or (N,>=6)")
        n, ld = d.shape[0], (d.shape[1] if d.shape[0] else 5)
        cap = n + 4096
        tl, sc, ids = np.empty((cap, 4), np.float64), np.empty(cap, np.float32), np.empty(cap, np.int64)
        m = C.c_int(0)
        rc = _lib().uni_byte_update(self._h, d.ctypes.data, n, ld, float(img_info[0]), float(img_info[1]), float(img_size[0]),
                                    float(img_size[1]), cap, tl.ctypes.data, sc.ctypes.data, ids.ctypes.data, C.byref(m))
        if rc != 0:
            raise UnicornAssocError(assoc_lib().uni_qd_last_error().decode())
        self.frame_id += 1
        return [STrack(tl[i].copy(), int(ids[i]), sc[i]) for i in range(m.value)]
