"""Golden vectors for the ByteTrack association (row N2): run the REAL reference classes
(/root/reference/unicorn/tracker/byte_tracker.py + matching.py + kalman_filter.py; lap / cython_bbox stubbed by
oracle/ref_bootstrap.py, see oracle/bytetrack_oracle.py) over oracle.bytetrack_oracle.synth_detections.
Run in the build container:  python tests/golden/make_golden_byte.py
"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_bootstrap  # noqa: E402

ref_bootstrap._install_stubs()
import torch  # noqa: E402,F401
import importlib.util  # noqa: E402

# load the tracker sub-package without executing unicorn/__init__.py (it pulls the whole model zoo)
pkg_u = types.ModuleType("unicorn"); pkg_u.__path__ = [os.path.join(ref_bootstrap.REF_ROOT, "unicorn")]
pkg_t = types.ModuleType("unicorn.tracker"); pkg_t.__path__ = [os.path.join(ref_bootstrap.REF_ROOT, "unicorn", "tracker")]
sys.modules["unicorn"], sys.modules["unicorn.tracker"] = pkg_u, pkg_t
for name in ("kalman_filter", "basetrack", "matching", "byte_tracker"):
    spec = importlib.util.spec_from_file_location("unicorn.tracker." + name, os.path.join(pkg_t.__path__[0], name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["unicorn.tracker." + name] = mod
    setattr(pkg_t, name, mod)
    spec.loader.exec_module(mod)
bt = sys.modules["unicorn.tracker.byte_tracker"]
import bytetrack_oracle as bo  # noqa: E402

out = {}
for name, kw, seed in [("default", dict(track_thresh=0.6, track_buffer=30, match_thresh=0.9, mot20=False), 0),
                       ("mot20", dict(track_thresh=0.5, track_buffer=10, match_thresh=0.8, mot20=True), 1),
                       ("crowded", dict(track_thresh=0.6, track_buffer=30, match_thresh=0.9, mot20=False), 2)]:
    sys.modules["unicorn.tracker.basetrack"].BaseTrack._count = 0
    trk = bt.BYTETracker(types.SimpleNamespace(**kw), frame_rate=30)
    frames, info, size = bo.synth_detections(seed=seed, n_obj=12 if name != "crowded" else 30)
    for f, d in enumerate(frames):
        res = trk.update(d.copy(), info, size)
        out["%s/%d/ids" % (name, f)] = np.array([t.track_id for t in res], dtype=np.int64)
        out["%s/%d/tlwh" % (name, f)] = np.array([t.tlwh for t in res], dtype=np.float64).reshape(-1, 4)
        out["%s/%d/score" % (name, f)] = np.array([t.score for t in res], dtype=np.float32)
    out["%s/count" % name] = np.array(sys.modules["unicorn.tracker.basetrack"].BaseTrack._count)
    out["%s/lost" % name] = np.array(sorted(t.track_id for t in trk.lost_stracks), dtype=np.int64)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "byte_sequence.npz"), **out)
print("wrote", len(out), "arrays;", {k: int(v) for k, v in out.items() if k.endswith("count")})
