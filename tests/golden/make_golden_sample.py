"""Golden vectors for the instance-embedding sampling, produced by EXECUTING the reference's own
source lines: unicorn/evaluators/mot_evaluator.py:1024-1034 are read from /root/reference, dedented and exec'd in a
namespace holding the names they use (bboxes, s, self.img_size, embed_cur, torch, F).  Run in the build container:

    python tests/golden/make_golden_sample.py        -> tests/golden/sample_embed_ref.npz
"""
import os
import textwrap
import types

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference/unicorn/evaluators/mot_evaluator.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    src = open(REF).read().split("\n")
    lines = src[1023:1034]                                    # 1-based 1024..1034
    assert "cx, cy = (bboxes[:, 0] + bboxes[:, 2])/2/s - 0.5" in lines[0], lines[0]
    assert "track_feats = torch.stack(track_feat_list, dim=0)" in lines[-1], lines[-1]
    code = "track_feat_list = []\n" + textwrap.dedent("\n".join(lines))
    out = {}
    for tag, (H, W, seed, C) in {"a": (320, 512, 0, 16), "b": (352, 608, 1, 8)}.items():      # small maps keep the fixture small
        g = torch.Generator().manual_seed(seed)
        s = 8
        embed_cur = torch.randn(1, C, H // s, W // s, generator=g)
        n = 48
        c = torch.rand(n, 2, generator=g) * torch.tensor([W + 40.0, H + 40.0]) - 20.0          # some centres outside the image
        wh = torch.rand(n, 2, generator=g) * 200 + 4
        bboxes = torch.cat([c - wh / 2, c + wh / 2], 1)
        bboxes[0] = torch.tensor([0.0, 0.0, 2.0, 2.0])                                           # corner cases
        bboxes[1] = torch.tensor([W - 2.0, H - 2.0, float(W), float(H)])
        ns = {"torch": torch, "F": F, "bboxes": bboxes, "s": s, "embed_cur": embed_cur,
              "self": types.SimpleNamespace(img_size=(H, W))}
        exec(code, ns)
        out["embed_" + tag] = embed_cur.numpy()
        out["boxes_" + tag] = bboxes.numpy()
        out["feats_" + tag] = ns["track_feats"].numpy()
    np.savez_compressed(os.path.join(HERE, "sample_embed_ref.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
