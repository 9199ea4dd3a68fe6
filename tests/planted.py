"""Planted head outputs for the post-processing parity tests (shared by the CPU / GPU tests and tests/golden/make_golden_post.py)."""
import torch


def planted_pred(A, nc, seed, n_clusters=40, per=12):
    """decoded head output (1, A, 5+nc) with clusters of overlapping high-score boxes (NMS has work to do), exact score
    ties and a low-score background"""
    n_clusters = min(n_clusters, A // (2 * per))
    g = torch.Generator().manual_seed(seed)
    pred = torch.zeros(1, A, 5 + nc)
    pred[0, :, 0] = torch.rand(A, generator=g) * 1280
    pred[0, :, 1] = torch.rand(A, generator=g) * 800
    pred[0, :, 2:4] = torch.rand(A, 2, generator=g) * 60 + 4
    pred[0, :, 4] = torch.rand(A, generator=g) * 0.05
    pred[0, :, 5:] = torch.rand(A, nc, generator=g) * 0.05
    idx = torch.randperm(A, generator=g)[:n_clusters * per].reshape(n_clusters, per)
    for c in range(n_clusters):
        cx, cy = torch.rand(2, generator=g) * torch.tensor([1200.0, 720.0]) + 40
        w, h = torch.rand(2, generator=g) * 120 + 30
        for j, a in enumerate(idx[c]):
            jit = (torch.rand(4, generator=g) - 0.5) * torch.tensor([0.5 * w, 0.5 * h, 0.3 * w, 0.3 * h])
            pred[0, a, :4] = torch.stack([cx, cy, w, h]) + jit
            pred[0, a, 4] = 0.5 + 0.5 * torch.rand(1, generator=g)
            pred[0, a, 5 + int(torch.randint(nc, (1,), generator=g))] = 0.6 + 0.4 * torch.rand(1, generator=g)
    pred[0, idx[0, 1], 4:] = pred[0, idx[0, 0], 4:]          # an exact score tie between two overlapping boxes
    return pred


def confident_head(P, obj_bias=1.0, cls_bias=1.0, spread=1.0):
    """Synthetic weights give obj * cls ~ 1e-4 (prediction biases at -4.6, exp/unicorn_track.py:146); loops whose logic runs on score thresholds
    (BYTETracker: track_thresh, the 0.1 low-score floor, det_thresh = track_thresh + 0.1) need detector-like scores.  Shift the obj / cls prediction
    biases so that scores sit around sigmoid(obj_bias) * sigmoid(cls_bias) and scale the obj / cls prediction WEIGHTS by `spread` (the
    feature-dependent part of the logits: a wider score distribution) -> new state dict."""
    Q = dict(P)
    for k in P:
        if k.startswith("head.") and (".obj_preds" in k or ".cls_preds" in k):
            if k.endswith(".bias"):
                Q[k] = torch.full_like(P[k], obj_bias if ".obj_preds" in k else cls_bias)
            elif k.endswith(".weight") and spread != 1.0:
                Q[k] = P[k] * spread
    return Q
