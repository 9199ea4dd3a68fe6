"""Golden vectors for the QuasiDense association (row N2): run the REAL reference class
(/root/reference/unicorn/tracker/quasi_dense_embed_tracker.py) over oracle.assoc_oracle.synth_sequence and record every
frame's outputs.  Run in the build container (the reference tree is absent on the GPU box):
    python tests/golden/make_golden_qd.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_bootstrap  # noqa: E402

ref_bootstrap._install_stubs()
import torch  # noqa: E402

sys.path.insert(0, ref_bootstrap.REF_ROOT)
import importlib.util  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_qd", os.path.join(ref_bootstrap.REF_ROOT, "unicorn/tracker/quasi_dense_embed_tracker.py"))
ref_qd = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_qd)
import assoc_oracle as ao  # noqa: E402

out = {}
for name, kw, seed in [("default", {}, 0), ("softmax_nocats", dict(match_metric="softmax", with_cats=False, memo_tracklet_frames=5), 1),
                       ("cosine", dict(match_metric="cosine", match_score_thr=0.6, memo_backdrop_frames=2), 2)]:
    trk = ref_qd.QuasiDenseEmbedTracker(**kw)
    frames = ao.synth_sequence(seed=seed)
    for f, (b, l, e) in enumerate(frames):
        rb, rl, ri, rv = trk.match(b.clone(), l.clone(), e.clone(), f, return_index=True)
        out["%s/%d/bboxes" % (name, f)] = rb.numpy()
        out["%s/%d/labels" % (name, f)] = rl.numpy()
        out["%s/%d/ids" % (name, f)] = ri.numpy()
        out["%s/%d/valids" % (name, f)] = rv.numpy()
    out["%s/num_tracklets" % name] = np.array(int(trk.num_tracklets))
    out["%s/alive" % name] = np.array(sorted(trk.tracklets.keys()), dtype=np.int64)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "qd_sequence.npz"), **out)
print("wrote", len(out), "arrays; tracklets:", {k: int(v) for k, v in out.items() if k.endswith("num_tracklets")})
