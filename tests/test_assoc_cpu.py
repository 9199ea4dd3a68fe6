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


# ---------------------------------------------------------------------------------------------------------------------
# ByteTrack
# ---------------------------------------------------------------------------------------------------------------------
import types  # noqa: E402

import bytetrack_oracle as bo  # noqa: E402
from unicorn_amd.tracker import byte_tracker as nbt  # noqa: E402

BGOLD = np.load(os.path.join(ROOT, "tests", "golden", "byte_sequence.npz"))
BCASES = [("default", dict(track_thresh=0.6, track_buffer=30, match_thresh=0.9, mot20=False), 0, 12),
          ("mot20", dict(track_thresh=0.5, track_buffer=10, match_thresh=0.8, mot20=True), 1, 12),
          ("crowded", dict(track_thresh=0.6, track_buffer=30, match_thresh=0.9, mot20=False), 2, 30)]


def test_byte_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "unicorn_assoc.h")).read()
    names = set(re.findall(r"\b(uni_byte_[a-z_]+)\s*\(", hdr))
    assert {"uni_byte_create", "uni_byte_destroy", "uni_byte_update", "uni_byte_id_count", "uni_byte_clean_id", "uni_byte_lost"} <= names
    for n in names:
        assert hasattr(assoc_lib(), n), n


@pytest.mark.parametrize("name,kw,seed,nobj", BCASES)
def test_byte_oracle_matches_reference_golden(name, kw, seed, nobj):
    st = bo.ByteState(**kw)
    frames, info, size = bo.synth_detections(seed=seed, n_obj=nobj)
    for f, d in enumerate(frames):
        res = bo.byte_update(st, d, info, size)
        assert np.array_equal(np.array([t.track_id for t in res], dtype=np.int64), BGOLD["%s/%d/ids" % (name, f)]), f
        assert np.allclose(np.array([t.tlwh for t in res]).reshape(-1, 4), BGOLD["%s/%d/tlwh" % (name, f)], rtol=0, atol=1e-9), f
        assert np.array_equal(np.array([t.score for t in res], dtype=np.float32), BGOLD["%s/%d/score" % (name, f)]), f
    assert st.count == int(BGOLD[name + "/count"])
    assert sorted(t.track_id for t in st.lost) == BGOLD[name + "/lost"].tolist()


@pytest.mark.parametrize("name,kw,seed,nobj", BCASES)
def test_byte_native_matches_reference_golden(name, kw, seed, nobj):
    nbt.clean_id()
    trk = nbt.BYTETracker(types.SimpleNamespace(**kw), frame_rate=30)
    frames, info, size = bo.synth_detections(seed=seed, n_obj=nobj)
    for f, d in enumerate(frames):
        res = trk.update(d.copy(), info, size)
        assert np.array_equal(np.array([t.track_id for t in res], dtype=np.int64), BGOLD["%s/%d/ids" % (name, f)]), f
        assert np.allclose(np.array([t.tlwh for t in res]).reshape(-1, 4), BGOLD["%s/%d/tlwh" % (name, f)], rtol=0, atol=1e-8), f
        assert np.array_equal(np.array([t.score for t in res], dtype=np.float32), BGOLD["%s/%d/score" % (name, f)]), f
    assert nbt.id_count() == int(BGOLD[name + "/count"])
    assert sorted(trk.lost_ids) == BGOLD[name + "/lost"].tolist()


@pytest.mark.parametrize("seed", [5, 6, 7, 8])
def test_byte_native_matches_oracle_more_sequences(seed):
    kw = dict(track_thresh=0.55, track_buffer=20, match_thresh=0.85, mot20=bool(seed & 1))
    nbt.clean_id()
    st, trk = bo.ByteState(**kw), nbt.BYTETracker(types.SimpleNamespace(**kw), frame_rate=25)
    st.max_time_lost = int(25 / 30.0 * 20)
    frames, info, size = bo.synth_detections(n_frames=80, n_obj=24, seed=seed)
    for f, d in enumerate(frames):
        det = d if f % 7 else np.concatenate([d, np.ones((len(d), 1), np.float32)], 1)      # (N,6) rows every 7th frame
        o = bo.byte_update(st, det, info, size)
        r = trk.update(det, info, size)
        assert [t.track_id for t in r] == [t.track_id for t in o], f
        assert np.allclose(np.array([t.tlwh for t in r]).reshape(-1, 4), np.array([t.tlwh for t in o]).reshape(-1, 4), rtol=0, atol=1e-8)
    assert nbt.id_count() == st.count


def test_byte_native_edges():
    nbt.clean_id()
    trk = nbt.BYTETracker(types.SimpleNamespace(track_thresh=0.6, track_buffer=30, match_thresh=0.9, mot20=False))
    assert trk.update(np.zeros((0, 5), np.float32), (1080, 1920), (800, 1440)) == []          # empty frame 1
    one = np.array([[100, 100, 200, 300, 0.9]], np.float32)
    assert trk.update(one, (1080, 1920), (800, 1440)) == []              # born after frame 1: unconfirmed until re-observed
    out = trk.update(one, (1080, 1920), (800, 1440))
    assert [t.track_id for t in out] == [1] and abs(out[0].tlwh[2] * out[0].tlwh[3] - (100 / 0.7407407) * (200 / 0.7407407)) < 60
    with pytest.raises(ValueError):
        trk.update(np.zeros((3, 4), np.float32), (1080, 1920), (800, 1440))


def test_native_non_finite_embeddings_are_deterministic():
    """a NaN / inf embedding row must not reach an int conversion (UB) in the vectorised exp: NaN similarities propagate
    like std::exp's (the detection matches nothing and starts / keeps its own track), the call succeeds, repeated runs agree."""
    def run():
        trk = QuasiDenseEmbedTracker()
        outs = []
        for f, (b, l, e) in enumerate(ao.synth_sequence(n_frames=6, n_obj=8, seed=11)):
            e = e.clone()
            if f >= 2:
                e[0] = float("nan")
                e[1, 3] = float("inf")
            outs.append(trk.match(b, l, e, f)[2].clone())
        return outs
    a, b = run(), run()
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert all(x.numel() > 0 for x in a)
