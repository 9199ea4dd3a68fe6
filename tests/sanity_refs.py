"""INDEPENDENT second statements of the three third-party operations the path depends on and whose real libraries are absent offline
(cv2.resize, torchvision.ops.nms / batched_nms, pycocotools rleFrString).  They do not make those rows "pinned" (only an output of the
real library could); they stop the oracle <-> kernel agreement from being self-referential: each one is written from the library's
DOCUMENTED behaviour by a different route than oracle/*.py (float arithmetic instead of OpenCV's fixed point, scalar O(N^2) loops
instead of vectorised suppression, a character-level decoder instead of the encoder's inverse), so a coordinate, rounding-direction
or tie-order mistake shared by oracle and kernel would show.  VERDICT r04 "next round" #6.  Test helpers only."""
import numpy as np
import torch
import torch.nn.functional as F


def float_letterbox(img, input_size, swap_rb):
    """PreprocessorX.process / preproc (unicorn_sot.py:111-123, data_augment.py:194-214) with the resize done by torch's FLOAT
    bilinear (`F.interpolate(align_corners=False)`, half-pixel centres like cv2.INTER_LINEAR, no antialiasing) on the uint8 image:
    -> ((3,H,W) float32 un-rounded, r, (rh, rw)); cv2's 11-bit fixed point may differ from round(.) of this by at most 1 LSB."""
    h, w = img.shape[:2]
    H, W = input_size
    r = min(H / h, W / w)
    rh, rw = int(h * r), int(w * r)
    src = img[:, :, ::-1] if swap_rb else img
    t = torch.from_numpy(np.ascontiguousarray(src)).permute(2, 0, 1)[None].double()
    rs = F.interpolate(t, size=(rh, rw), mode="bilinear", align_corners=False)[0]
    out = torch.full((3, H, W), 114.0, dtype=torch.float64)
    out[:, :rh, :rw] = rs
    return out.numpy(), r, (rh, rw)


def brute_nms(boxes, scores, thr):
    """torchvision.ops.nms from its docstring: "iteratively removes lower scoring boxes which have an IoU greater than iou_threshold
    with another (higher scoring) box"; the result is the kept indices "sorted in decreasing order of scores".  Scalar loops, fp32
    arithmetic per element (numpy float32 scalars), ties in score resolved by the lower index first."""
    b = np.asarray(boxes, dtype=np.float32)
    s = np.asarray(scores, dtype=np.float32)
    n = b.shape[0]
    order = sorted(range(n), key=lambda i: (-float(s[i]), i))
    keep = []
    for i in order:
        ok = True
        for j in keep:
            xx1, yy1 = max(b[i, 0], b[j, 0]), max(b[i, 1], b[j, 1])
            xx2, yy2 = min(b[i, 2], b[j, 2]), min(b[i, 3], b[j, 3])
            iw, ih = np.float32(max(np.float32(xx2 - xx1), np.float32(0))), np.float32(max(np.float32(yy2 - yy1), np.float32(0)))
            inter = np.float32(iw * ih)
            ai = np.float32(np.float32(b[i, 2] - b[i, 0]) * np.float32(b[i, 3] - b[i, 1]))
            aj = np.float32(np.float32(b[j, 2] - b[j, 0]) * np.float32(b[j, 3] - b[j, 1]))
            iou = np.float32(inter / np.float32(np.float32(aj + ai) - inter))
            if iou > np.float32(thr):
                ok = False
                break
        if ok:
            keep.append(i)
    return keep


def brute_batched_nms(boxes, scores, idxs, thr):
    """torchvision.ops.batched_nms: "each index value correspond to a category, and NMS will not be applied between elements of
    different categories" -- done LITERALLY per category (no coordinate offset), merged by descending score (ties: lower index)."""
    b, s, c = np.asarray(boxes, np.float32), np.asarray(scores, np.float32), np.asarray(idxs)
    keep = []
    for k in np.unique(c):
        sel = np.nonzero(c == k)[0]
        keep += [int(sel[i]) for i in brute_nms(b[sel], s[sel], thr)]
    return sorted(keep, key=lambda i: (-float(s[i]), i))


def coco_rle_string_to_mask(s, h, w):
    """pycocotools `rleFrString` + `rleDecode` as DOCUMENTED in maskApi.c/.h: the string is a sequence of counts, each written
    low-to-high in 5-bit groups as chars `48 + (bits | 0x20 if more groups follow)`; bit 0x10 of the LAST group is the sign
    (extended upwards); from the FOURTH count on (`i > 2` in rleToString / `m > 2` in rleFrString) a count is stored as the difference
    to the count two places earlier; counts are run
    lengths of 0s and 1s alternating, starting with 0s, over the mask in COLUMN-major order.  Character-level state machine."""
    if isinstance(s, bytes):
        s = s.decode("ascii")
    counts, p = [], 0
    while p < len(s):
        x, k = 0, 0
        while True:
            c = ord(s[p]) - 48
            p += 1
            x |= (c & 0x1F) << (5 * k)
            k += 1
            if not (c & 0x20):
                if c & 0x10:
                    x -= 1 << (5 * k)
                break
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    flat = np.zeros(h * w, dtype=np.uint8)
    pos, v = 0, 0
    for c in counts:
        assert c >= 0 and pos + c <= h * w, "run past the mask"
        if v:
            flat[pos:pos + c] = 1
        pos += c
        v ^= 1
    assert pos == h * w, "runs do not cover the mask"
    return flat.reshape(w, h).T.copy()


def adversarial_nms_cases():
    """-> list of (name, boxes (N,4) fp32 xyxy, scores (N,), class idx (N,), threshold)"""
    cases = []
    # IoU EXACTLY at the threshold (inter 2 / union 4 = 0.5; every quantity exact in fp32): `>` must NOT suppress; a hair more must
    b = np.array([[0, 0, 2, 2], [0, 0, 2, 1], [10, 10, 12, 12], [10, 10, 12, 11.000001], [20, 20, 24, 24], [20, 20, 24, 22.5]], np.float32)
    cases.append(("iou_at_threshold", b, np.array([.9, .8, .7, .6, .5, .4], np.float32), np.zeros(6, np.int64), 0.5))
    # equal scores: identical boxes, chains a-b-c where b overlaps both and a, c do not overlap each other
    b = np.array([[0, 0, 10, 10], [0, 0, 10, 10], [6, 0, 16, 10], [12, 0, 22, 10], [0, 0, 10, 10], [40, 40, 50, 50], [41, 41, 51, 51]], np.float32)
    cases.append(("equal_scores", b, np.array([.5, .5, .5, .5, .7, .5, .5], np.float32), np.zeros(7, np.int64), 0.3))
    # class offsets near the fp32 resolution: coordinates ~4e3 and 80 classes put the coordinate-trick offsets at ~3e5 (ulp 1/32):
    # boxes a quarter pixel apart, same box in different classes (must all survive), near-duplicates inside a class (must not)
    g = np.random.default_rng(0)
    xy = g.uniform(3000, 4000, (60, 2)).astype(np.float32)
    wh = g.uniform(8, 64, (60, 2)).astype(np.float32)
    base = np.concatenate([xy, xy + wh], 1)
    b = np.concatenate([base, base + np.float32(0.25), base], 0).astype(np.float32)
    cl = np.concatenate([g.integers(0, 80, 60), np.zeros(60, np.int64), np.zeros(60, np.int64)])
    cl[60:120] = cl[:60]                     # the shifted copies share the class of their original (near-duplicates, IoU ~0.95)
    cl[120:] = (cl[:60] + 1) % 80            # exact copies in the NEXT class: survive
    sc = g.uniform(0.1, 1.0, 180).astype(np.float32)
    cases.append(("class_offsets_fp32", b, sc, cl.astype(np.int64), 0.65))
    # a dense random field with many near-threshold pairs
    xy = g.uniform(0, 200, (300, 2)).astype(np.float32)
    wh = g.uniform(20, 60, (300, 2)).astype(np.float32)
    cases.append(("dense", np.concatenate([xy, xy + wh], 1), np.round(g.uniform(0, 1, 300), 2).astype(np.float32), g.integers(0, 3, 300), 0.45))
    return cases


def rle_sanity_masks():
    """1-pixel, all-ones, column-alternating (every run has length h), row-alternating (runs of 1) and an empty mask"""
    out = []
    for h, w in [(1, 1), (5, 7), (33, 65)]:
        one = np.zeros((h, w), np.uint8)
        one[h // 2, w // 2] = 1
        first = np.zeros((h, w), np.uint8)
        first[0, 0] = 1
        last = np.zeros((h, w), np.uint8)
        last[-1, -1] = 1
        cols = np.zeros((h, w), np.uint8)
        cols[:, ::2] = 1
        rows = np.zeros((h, w), np.uint8)
        rows[1::2, :] = 1
        out += [("one_pixel", one), ("first_pixel", first), ("last_pixel", last), ("all_ones", np.ones((h, w), np.uint8)),
                ("column_alternating", cols), ("row_alternating", rows), ("empty", np.zeros((h, w), np.uint8))]
    return out
