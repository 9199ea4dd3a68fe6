"""Row N2 (association): the oracle restatement is pinned to golden vectors produced by the REAL reference class, and the
native library (include/unicorn_assoc.h) is held to the golden and to the oracle on further seeds.  CPU only."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import assoc_oracle as ao  # noqa: E402

from unicorn_amd.tracker.quasi_dense_embed_tracker import QuasiDenseEmbedTracker, assoc_lib  # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "qd_sequence.npz"))
CASES = [("default", {}, 0), ("softmax_nocats", dict(match_metric="softmax", with_cats=False, memo_tracklet_frames=5), 1),
         ("cosine", dict(match_metric="cosine", match_score_thr=0.6, memo_backdrop_frames=2), 2)]


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "unicorn_assoc.h")).read()
    names = set(re.findall(r"\b(uni_qd_[a-z_]+)\s*\(", hdr))
    assert {"uni_qd_create", "uni_qd_destroy", "uni_qd_match", "uni_qd_default_cfg", "uni_qd_num_tracklets", "uni_qd_alive",
            "uni_qd_last_error"} <= names
    lib = assoc_lib()
    for n in names:
        assert hasattr(lib, n), n


@pytest.mark.parametrize("name,kw,seed", CASES)
def test_oracle_matches_reference_golden(name, kw, seed):
    st = ao.QDState(**kw)
    for f, (b, l, e) in enumerate(ao.synth_sequence(seed=seed)):
        rb, rl, ri, rv = ao.qd_match(st, b, l, e, f)
        assert np.array_equal(ri.numpy(), GOLD["%s/%d/ids" % (name, f)]), f
        assert np.array_equal(rv.numpy(), GOLD["%s/%d/valids" % (name, f)]), f
        assert np.array_equal(rl.numpy(), GOLD["%s/%d/labels" % (name, f)]), f
        assert np.array_equal(rb.numpy(), GOLD["%s/%d/bboxes" % (name, f)]), f
    assert st.num_tracklets == int(GOLD[name + "/num_tracklets"])
    assert sorted(st.tracklets.keys()) == GOLD[name + "/alive"].tolist()


@pytest.mark.parametrize("name,kw,seed", CASES)
def test_native_matches_reference_golden(name, kw, seed):
    trk = QuasiDenseEmbedTracker(**kw)
    assert trk.empty
    for f, (b, l, e) in enumerate(ao.synth_sequence(seed=seed)):
        rb, rl, ri, rv = trk.match(b, l, e, f, return_index=True)
        assert np.array_equal(ri.numpy(), GOLD["%s/%d/ids" % (name, f)]), f
        assert np.array_equal(rv.numpy(), GOLD["%s/%d/valids" % (name, f)]), f
        assert np.array_equal(rl.numpy(), GOLD["%s/%d/labels" % (name, f)]), f
        assert np.array_equal(rb.numpy(), GOLD["%s/%d/bboxes" % (name, f)]), f
    assert trk.num_tracklets == int(GOLD[name + "/num_tracklets"])
    assert sorted(trk.tracklet_ids) == GOLD[name + "/alive"].tolist()


@pytest.mark.parametrize("seed", [3, 4, 5, 6])
@pytest.mark.parametrize("kw", [{}, dict(match_metric="softmax"), dict(with_cats=False, memo_momentum=0.5, init_score_thr=0.6)])
def test_native_matches_oracle_more_sequences(seed, kw):
    st, trk = ao.QDState(**kw), QuasiDenseEmbedTracker(**kw)
    for f, (b, l, e) in enumerate(ao.synth_sequence(n_frames=30, n_obj=22, seed=seed, classes=2)):
        ob, ol, oi, ov = ao.qd_match(st, b, l, e, f)
        nb, nl, ni = trk.match(b, l, e, f)
        assert torch.equal(ni, oi) and torch.equal(nl, ol) and torch.equal(nb, ob), f
    assert trk.num_tracklets == st.num_tracklets and trk.tracklet_ids == list(st.tracklets.keys())


def test_native_errors_and_edges():
    trk = QuasiDenseEmbedTracker()
    b, l, e = torch.zeros(0, 5), torch.zeros(0, dtype=torch.long), torch.zeros(0, 128)
    rb, rl, ri = trk.match(b, l, e, 0)                       # empty frame before anything was seen
    assert rb.shape == (0, 5) and ri.numel() == 0 and trk.empty
    one = torch.tensor([[10.0, 10, 50, 60, 0.9]])
    assert trk.match(one, torch.tensor([1]), torch.ones(1, 16), 1)[2].tolist() == [0]
    with pytest.raises(RuntimeError):
        trk.match(one, torch.tensor([1]), torch.ones(1, 32), 2)       # embedding dim changed
    with pytest.raises(ValueError):
        trk.match(torch.zeros(2, 4), torch.zeros(2, dtype=torch.long), torch.zeros(2, 16), 3)
    with pytest.raises(AssertionError):
        QuasiDenseEmbedTracker(match_metric="l2")
