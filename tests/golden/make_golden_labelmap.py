"""Golden label maps of the SOT / VOS initialisation, produced by EXECUTING the reference's own `get_label_map`
(external/lib/test/tracker/unicorn_sot.py:128-139, `.cuda()` dropped) followed by the driver's 1/8 bilinear down-sampling
(:52-53) on boxes that exercise torch.round's half-to-even, clipping at the image border and empty / inverted boxes.
Run in the build container:

    python tests/golden/make_golden_labelmap.py        -> tests/golden/labelmap_ref.npz
"""
import os
import textwrap

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference/external/lib/test/tracker/unicorn_sot.py"
HERE = os.path.dirname(os.path.abspath(__file__))

BOXES = [[320.0, 200.0, 640.0, 400.0], [3.4, 7.5, 1279.6, 700.5], [-20.0, -5.0, 50.5, 2000.0], [100.5, 100.5, 100.5, 300.0],
         [0.5, 1.5, 2.5, 3.5], [10.0, 10.0, 5.0, 5.0], [1279.5, 799.5, 1400.0, 900.0], [8.0, 8.0, 16.0, 16.0], [7.49, 7.51, 16.5, 17.5],
         [-1e4, -1e4, 1e4, 1e4], [639.5, 0.0, 640.5, 800.0]]


def main():
    src = open(REF).read().split("\n")
    lines = src[127:139]                                      # 1-based 128..139
    assert lines[0].startswith("def get_label_map(boxes, H, W):") and "return labels" in lines[-1], (lines[0], lines[-1])
    code = textwrap.dedent("\n".join(lines)).replace(".cuda()", "")
    ns = {"torch": torch}
    exec(code, ns)
    out = {"boxes": np.array(BOXES, dtype=np.float32)}
    for tag, (H, W) in {"a": (800, 1280), "b": (320, 512)}.items():
        maps = []
        for b in BOXES:
            lab = ns["get_label_map"](torch.tensor(b), H, W)                                       # unicorn_sot.py:52
            maps.append(F.interpolate(lab, scale_factor=1 / 8, mode="bilinear", align_corners=False)[0].flatten(-2).numpy())   # :53
        out["hw_" + tag] = np.array([H, W])
        out["lbs_" + tag] = np.concatenate(maps, 0)
    np.savez_compressed(os.path.join(HERE, "labelmap_ref.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
