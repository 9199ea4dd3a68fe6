"""Association step alone (CPU): native library vs the python/torch restatement of the reference, same detection stream."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import assoc_oracle as ao
from unicorn_amd.tracker import QuasiDenseEmbedTracker
torch.set_num_threads(1)
frames = ao.synth_sequence(n_frames=200, n_obj=60, seed=0, classes=4)
for name, mk, call in [("oracle (torch, 1 thread)", lambda: ao.QDState(), lambda s, b, l, e, f: ao.qd_match(s, b, l, e, f)),
                       ("native (libunicorn_assoc)", lambda: QuasiDenseEmbedTracker(), lambda s, b, l, e, f: s.match(b, l, e, f))]:
    st = mk()
    t0 = time.perf_counter()
    for f, (b, l, e) in enumerate(frames):
        call(st, b, l, e, f)
    dt = time.perf_counter() - t0
    print("%-28s %7.2f ms/frame  (%d frames, ~%d detections/frame)" % (name, dt / len(frames) * 1e3, len(frames), sum(x[0].shape[0] for x in frames) // len(frames)))

# ---- ByteTrack: native vs the numpy restatement of the reference (same float64 arithmetic)
import types
import numpy as np
import bytetrack_oracle as bo
from unicorn_amd.tracker import byte_tracker as nbt
frames, info, size = bo.synth_detections(n_frames=300, n_obj=60, seed=0)
kw = dict(track_thresh=0.6, track_buffer=30, match_thresh=0.9, mot20=False)
for name, mk, call in [("oracle (numpy + scipy LAP)", lambda: bo.ByteState(**kw), lambda s, d: bo.byte_update(s, d, info, size)),
                       ("native (libunicorn_assoc)", lambda: nbt.BYTETracker(types.SimpleNamespace(**kw)), lambda s, d: s.update(d, info, size))]:
    nbt.clean_id()
    st = mk()
    t0 = time.perf_counter()
    for d in frames:
        call(st, d)
    dt = time.perf_counter() - t0
    print("ByteTrack %-28s %7.2f ms/frame  (%d frames, ~%d detections/frame)" % (name, dt / len(frames) * 1e3, len(frames), sum(len(x) for x in frames) // len(frames)))
