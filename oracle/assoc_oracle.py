"""CPU restatement of the reference's QuasiDense embedding association (row N2 of SURVEY.md §8f).

TEST INFRASTRUCTURE ONLY (see oracle/unicorn_oracle.py): only tests/ may import it; the product is the C++ library behind
include/unicorn_assoc.h.  Follows unicorn/tracker/quasi_dense_embed_tracker.py line by line (cited per function) as a plain
functional state machine over python lists (no torch modules), so the native implementation can be compared step by step.
Pinned: tests/golden/qd_sequence.npz was produced by the REAL reference class (tests/golden/make_golden_qd.py, run in the
build container) and tests/test_assoc_cpu.py holds this restatement to it exactly.
torchvision.ops.box_iou is third-party (absent offline): restated from its published semantics (see box_iou below).
"""
from typing import List, Tuple

import torch
from torch import Tensor


def box_iou(a: Tensor, b: Tensor) -> Tensor:
    """torchvision.ops.box_iou: inter / (area_a + area_b - inter), xyxy boxes."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, None, :2], b[None, :, :2])
    rb = torch.min(a[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area_a[:, None] + area_b[None] - inter)


class QDState:
    """quasi_dense_embed_tracker.py:11-42 (constructor defaults identical)"""

    def __init__(self, init_score_thr=0.8, obj_score_thr=0.5, match_score_thr=0.5, memo_tracklet_frames=30,
                 memo_backdrop_frames=1, memo_momentum=0.8, nms_conf_thr=0.5, nms_backdrop_iou_thr=0.3,
                 nms_class_iou_thr=0.7, with_cats=True, match_metric="bisoftmax"):
        self.init_score_thr, self.obj_score_thr, self.match_score_thr = init_score_thr, obj_score_thr, match_score_thr
        self.memo_tracklet_frames, self.memo_backdrop_frames = memo_tracklet_frames, memo_backdrop_frames
        self.memo_momentum, self.nms_conf_thr = memo_momentum, nms_conf_thr
        self.nms_backdrop_iou_thr, self.nms_class_iou_thr = nms_backdrop_iou_thr, nms_class_iou_thr
        self.with_cats, self.match_metric = with_cats, match_metric
        self.num_tracklets = 0
        self.tracklets = {}        # id -> dict(bbox, embed, label, last_frame, velocity, acc_frame); insertion ordered
        self.backdrops = []        # newest first


def _update_memo(st: QDState, ids, bboxes, embeds, labels, frame_id):
    """quasi_dense_embed_tracker.py:48-102"""
    for k in torch.nonzero(ids > -1).flatten().tolist():
        tid, bbox, embed, label = int(ids[k]), bboxes[k], embeds[k], labels[k]
        if tid in st.tracklets:
            t = st.tracklets[tid]
            velocity = (bbox - t["bbox"]) / (frame_id - t["last_frame"])
            t["bbox"] = bbox
            t["embed"] = (1 - st.memo_momentum) * t["embed"] + st.memo_momentum * embed
            t["last_frame"] = frame_id
            t["label"] = label
            t["velocity"] = (t["velocity"] * t["acc_frame"] + velocity) / (t["acc_frame"] + 1)
            t["acc_frame"] += 1
        else:
            st.tracklets[tid] = dict(bbox=bbox, embed=embed, label=label, last_frame=frame_id,
                                     velocity=torch.zeros_like(bbox), acc_frame=0)
    backdrop_inds = torch.nonzero(ids == -1, as_tuple=False).squeeze(1)
    ious = box_iou(bboxes[backdrop_inds, :-1], bboxes[:, :-1])
    for i, ind in enumerate(backdrop_inds.tolist()):
        if (ious[i, :ind] > st.nms_backdrop_iou_thr).any():
            backdrop_inds[i] = -1
    backdrop_inds = backdrop_inds[backdrop_inds > -1]
    st.backdrops.insert(0, dict(bboxes=bboxes[backdrop_inds], embeds=embeds[backdrop_inds], labels=labels[backdrop_inds]))
    for k in [k for k, v in st.tracklets.items() if frame_id - v["last_frame"] >= st.memo_tracklet_frames]:
        st.tracklets.pop(k)
    if len(st.backdrops) > st.memo_backdrop_frames:
        st.backdrops.pop()


def _memo(st: QDState):
    """quasi_dense_embed_tracker.py:104-135"""
    mb = [v["bbox"][None] for v in st.tracklets.values()]
    me = [v["embed"][None] for v in st.tracklets.values()]
    ml = [v["label"].view(1, 1) for v in st.tracklets.values()]
    ids = torch.tensor(list(st.tracklets.keys()), dtype=torch.long).view(1, -1)
    for b in st.backdrops:
        ids = torch.cat([ids, torch.full((1, b["embeds"].size(0)), -1, dtype=torch.long)], 1)
        mb.append(b["bboxes"]); me.append(b["embeds"]); ml.append(b["labels"][:, None])
    return torch.cat(mb, 0), torch.cat(ml, 0).squeeze(1), torch.cat(me, 0), ids.squeeze(0)


def qd_match(st: QDState, bboxes: Tensor, labels: Tensor, track_feats: Tensor, frame_id: int) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """quasi_dense_embed_tracker.py:137-212 -> (bboxes, labels, ids, valids)"""
    _, inds = bboxes[:, -1].sort(descending=True)
    bboxes, labels, embeds = bboxes[inds, :], labels[inds], track_feats[inds, :]
    valids = bboxes.new_ones((bboxes.size(0)))
    ious = box_iou(bboxes[:, :-1], bboxes[:, :-1])
    for i in range(1, bboxes.size(0)):
        thr = st.nms_backdrop_iou_thr if bboxes[i, -1] < st.obj_score_thr else st.nms_class_iou_thr
        if (ious[i, :i] > thr).any():
            valids[i] = 0
    valids = valids == 1
    bboxes, labels, embeds = bboxes[valids, :], labels[valids], embeds[valids, :]
    ids = torch.full((bboxes.size(0),), -1, dtype=torch.long)
    if bboxes.size(0) > 0 and st.tracklets:
        memo_bboxes, memo_labels, memo_embeds, memo_ids = _memo(st)
        if st.match_metric == "bisoftmax":
            feats = torch.mm(embeds, memo_embeds.t())
            scores = (feats.softmax(dim=1) + feats.softmax(dim=0)) / 2
        elif st.match_metric == "softmax":
            scores = torch.mm(embeds, memo_embeds.t()).softmax(dim=1)
        else:
            scores = torch.mm(torch.nn.functional.normalize(embeds, p=2, dim=1),
                              torch.nn.functional.normalize(memo_embeds, p=2, dim=1).t())
        if st.with_cats:
            scores = scores * (labels.view(-1, 1) == memo_labels.view(1, -1)).float()
        for i in range(bboxes.size(0)):
            conf, memo_ind = torch.max(scores[i, :], dim=0)
            tid = memo_ids[memo_ind]
            if conf > st.match_score_thr:
                if tid > -1:
                    if bboxes[i, -1] > st.obj_score_thr:
                        ids[i] = tid
                        scores[:i, memo_ind] = 0
                        scores[i + 1:, memo_ind] = 0
                    elif conf > st.nms_conf_thr:
                        ids[i] = -2
    new_inds = (ids == -1) & (bboxes[:, 4] > st.init_score_thr)
    num_news = int(new_inds.sum())
    ids[new_inds] = torch.arange(st.num_tracklets, st.num_tracklets + num_news, dtype=torch.long)
    st.num_tracklets += num_news
    _update_memo(st, ids, bboxes, embeds, labels, frame_id)
    return bboxes, labels, ids, valids


def synth_sequence(n_frames=40, n_obj=14, dim=128, seed=0, classes=3):
    """Deterministic detection stream for association tests: objects drift, get occluded (missed), produce duplicate and
    low-score detections (backdrops), change score; embeddings = per-object prototype + noise."""
    g = torch.Generator().manual_seed(seed)
    proto = torch.randn(n_obj, dim, generator=g) * 1.2
    pos = torch.rand(n_obj, 2, generator=g) * torch.tensor([1100.0, 650.0]) + 50
    size = torch.rand(n_obj, 2, generator=g) * 90 + 40
    vel = (torch.rand(n_obj, 2, generator=g) - 0.5) * 14
    cls = torch.randint(0, classes, (n_obj,), generator=g)
    frames = []
    for f in range(n_frames):
        pos = pos + vel + (torch.rand(n_obj, 2, generator=g) - 0.5) * 3
        rows, labs, feats = [], [], []
        for o in range(n_obj):
            if o >= 4 + f // 2 and o >= 6:          # objects enter over time
                continue
            if torch.rand(1, generator=g).item() < 0.12:      # missed detection
                continue
            sc = 0.3 + 0.7 * torch.rand(1, generator=g).item()
            if torch.rand(1, generator=g).item() < 0.15:
                sc = 0.2 + 0.25 * torch.rand(1, generator=g).item()       # low score -> backdrop candidate
            box = torch.cat([pos[o] - size[o] / 2, pos[o] + size[o] / 2, torch.tensor([sc])])
            rows.append(box); labs.append(cls[o]); feats.append(proto[o] + 0.35 * torch.randn(dim, generator=g))
            if torch.rand(1, generator=g).item() < 0.2:       # duplicate, slightly shifted, lower score
                d = box.clone(); d[:4] += (torch.rand(4, generator=g) - 0.5) * 10; d[4] = sc * 0.8
                rows.append(d); labs.append(cls[o] if torch.rand(1, generator=g).item() < 0.7 else (cls[o] + 1) % classes)
                feats.append(proto[o] + 0.5 * torch.randn(dim, generator=g))
        if f == 17:                                   # an empty frame
            rows, labs, feats = [], [], []
        if rows:
            frames.append((torch.stack(rows).float(), torch.stack(labs).long(), torch.stack(feats).float()))
        else:
            frames.append((torch.zeros(0, 5), torch.zeros(0, dtype=torch.long), torch.zeros(0, dim)))
    return frames
