"""Golden vectors for post-processing (row N1): the REAL reference `postprocess` (unicorn/utils/boxes.py:33-77; torchvision nms /
batched_nms stubbed by oracle/ref_bootstrap.py from their published semantics) on planted head outputs.
Run in the build container:  python tests/golden/make_golden_post.py"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_bootstrap  # noqa: E402

ref_bootstrap._install_stubs()
import torch  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_boxes", os.path.join(ref_bootstrap.REF_ROOT, "unicorn/utils/boxes.py"))
ref_boxes = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_boxes)
from planted import planted_pred  # noqa: E402

out = {}
for A, nc, agn in [(2100, 1, False), (5000, 8, False), (5000, 8, True), (333, 3, False)]:
    pred = planted_pred(A, nc, seed=A + nc)
    res = ref_boxes.postprocess(pred.clone(), nc, 0.2, 0.45, class_agnostic=agn)[0]
    out["%d_%d_%d" % (A, nc, int(agn))] = res.numpy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "postprocess_planted.npz"), **out)
print({k: v.shape for k, v in out.items()})
