"""Golden vectors for the mask post-processing of the VOS / MOTS drivers, produced by EXECUTING the
reference's own source lines (read from /root/reference, dedented, exec'd in a namespace that holds the names they use):

  external/lib/test/tracker/unicorn_vos.py:99-120      overlap handling + soft aggregation + final id map
  unicorn/evaluators/mot_evaluator.py:804-805          masks = F.interpolate(..., 1/scale)[:, 0, :img_h, :img_w] > mask_thres
  unicorn/evaluators/mot_evaluator.py:860-865          overlap-free masks (earlier tracks win)

Run in the build container:

    python tests/golden/make_golden_vos.py        -> tests/golden/vos_mots_ref.npz
"""
import copy
import os
import textwrap
import types

import numpy as np
import torch
import torch.nn.functional as F

VOS = "/root/reference/external/lib/test/tracker/unicorn_vos.py"
MOT = "/root/reference/unicorn/evaluators/mot_evaluator.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def vos_cases():
    """(tag, H, W, tracked ids, ids of objects introduced mid-sequence (groups), ids introduced in THIS frame)"""
    return [("a", 40, 56, ["2", "5"], [["3"]], ["7"]),
            ("b", 33, 47, ["1"], [], []),
            ("c", 64, 96, ["1", "2", "3", "4"], [["6", "9"], ["5"]], []),
            ("d", 24, 24, ["3"], [], ["1", "2"])]


def main():
    out = {}
    # ---- unicorn_vos.py:99-120 -----------------------------------------------------------------------------------------
    src = open(VOS).read().split("\n")
    lines = src[98:120]                                        # 1-based 99..120
    assert "# Deal with overlapped masks" in lines[0], lines[0]
    assert "final_mask[final_mask_dict[obj_id]==1] = int(obj_id)" in lines[-1], lines[-1]
    code = textwrap.dedent("\n".join(lines))
    for tag, H, W, init_ids, new_groups, frame_ids in vos_cases():
        g = np.random.default_rng(len(tag) + H)
        tracked = list(init_ids) + [i for grp in new_groups for i in grp]
        probs = g.random((len(tracked), H, W), dtype=np.float32)
        probs[0, :5] = 0.0                                     # rows where an object is certainly absent / present
        probs[-1, 5:9] = 1.0
        probs[:, -3:, -3:] = 0.0                               # a corner that must go to the background
        if len(tracked) > 1:
            probs[1, 10:14] = probs[0, 10:14]                  # exact ties between two objects: np.argmax keeps the lower id
        final_mask_dict = {k: probs[i] for i, k in enumerate(tracked)}
        info = {}
        if frame_ids:
            init_mask = np.zeros((H, W), dtype=np.uint8)
            for j, k in enumerate(frame_ids):
                init_mask[2 + 6 * j:7 + 6 * j, 3:15] = int(k)
            info = {"init_object_ids": list(frame_ids), "init_mask": init_mask}
            for k in frame_ids:                                # unicorn_vos.py:98
                final_mask_dict[k] = (init_mask == int(k))
        ns = {"np": np, "copy": copy, "info": info, "final_mask_dict": final_mask_dict,
              "self": types.SimpleNamespace(H=H, W=W, init_object_ids=list(init_ids), obj_ids_new=[list(x) for x in new_groups])}
        exec(code, ns)
        out["vos_%s_probs" % tag] = probs
        out["vos_%s_ids" % tag] = np.array([int(k) for k in tracked], dtype=np.int64)
        out["vos_%s_init" % tag] = np.stack([(info["init_mask"] == int(k)) for k in frame_ids]).astype(np.uint8) \
            if frame_ids else np.zeros((0, H, W), np.uint8)
        out["vos_%s_init_ids" % tag] = np.array([int(k) for k in frame_ids], dtype=np.int64)
        out["vos_%s_final" % tag] = ns["final_mask"]
        assert ns["cur_obj_ids"] == tracked + list(frame_ids)
    # ---- mot_evaluator.py:804-805 + 860-865 -----------------------------------------------------------------------------
    src = open(MOT).read().split("\n")
    l_rs = src[803:805]
    assert "masks = F.interpolate(outputs_mask[0], scale_factor=1/scale" in l_rs[0] and "> self.mask_thres" in l_rs[1], l_rs
    l_of = src[859:865]
    assert "if masks.size(0) > 0:" in l_of[0] and "mask_prev = torch.logical_or(mask_prev, masks[n])" in l_of[-1], l_of
    code_rs, code_of = textwrap.dedent("\n".join(l_rs)), textwrap.dedent("\n".join(l_of))
    for tag, (N, Hn, Wn, img_h, img_w, seed) in {"a": (5, 80, 128, 135, 240, 0), "b": (1, 96, 160, 110, 190, 1),
                                                  "c": (9, 100, 160, 300, 480, 2), "d": (3, 64, 64, 64, 64, 3)}.items():
        gen = torch.Generator().manual_seed(seed)
        scale = min(Hn / float(img_h), Wn / float(img_w))
        blobs = torch.rand(N, 1, Hn // 8, Wn // 8, generator=gen)
        prob = F.interpolate(blobs, size=(Hn, Wn), mode="bilinear", align_corners=False)     # smooth maps that cross 0.5
        prob = (prob + 0.05 * torch.rand(N, 1, Hn, Wn, generator=gen)).clamp(0, 1)
        ns = {"F": F, "torch": torch, "outputs_mask": [prob], "scale": scale, "img_h": img_h, "img_w": img_w,
              "self": types.SimpleNamespace(mask_thres=0.5)}
        exec(code_rs, ns)
        masks = ns["masks"]
        ns2 = {"torch": torch, "masks": masks}
        exec(code_of, ns2)
        out["mots_%s_prob" % tag] = prob[:, 0].numpy()
        out["mots_%s_geom" % tag] = np.array([scale, img_h, img_w], dtype=np.float64)
        out["mots_%s_masks" % tag] = np.packbits(masks.numpy().astype(np.uint8).reshape(-1))
        out["mots_%s_free" % tag] = np.packbits(ns2["masks_new"].numpy().astype(np.uint8).reshape(-1))
        out["mots_%s_shape" % tag] = np.array(masks.shape, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "vos_mots_ref.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
