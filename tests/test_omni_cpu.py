"""Host logic of unicorn_amd/tracker/omni.py that needs no GPU: which frame is the interaction reference of which
(unicorn/evaluators/mot_evaluator.py:1005-1020: the interaction runs, and `pre_dict` advances, only on frames WITH detections; the
first such frame is its own reference)."""
import torch

from unicorn_amd.tracker.omni import OmniMOTFrame


class _FakeModel:
    def __init__(self):
        self.calls = []

    def __call__(self, seq_dict0=None, seq_dict1=None, feat=None, mode=None):
        if mode == "interaction":
            self.calls.append((seq_dict0["feat"].clone(), seq_dict1["feat"].clone()))
            return seq_dict0["feat"], seq_dict1["feat"]
        return feat                                   # upsample


def _ticket(vals):
    from types import SimpleNamespace
    feat = torch.tensor(vals, dtype=torch.float32).view(-1, 1, 1, 1).expand(-1, 2, 3, 4).contiguous()
    return SimpleNamespace(B=len(vals), cur={"feat": feat, "pos": torch.zeros(len(vals), 2, 3, 4), "h": 3, "w": 4})


def _refs(model):
    return [float(v) for v in model.calls[-1][0][:, 0, 0, 0]]


def test_reference_frame_is_the_last_frame_with_detections():
    m = _FakeModel()
    o = OmniMOTFrame(m, None, (32, 32))
    # batch of 4 frames, frame 0 has no detections: frame 1 is its own reference, frame 2 (none) is skipped, frame 3 refers to frame 1
    o._interact(_ticket([10., 11., 12., 13.]), [False, True, False, True])
    r = _refs(m)
    assert r[1] == 11. and r[3] == 11. and float(o.pre_dict["feat"][0, 0, 0, 0]) == 13.
    # next call (one frame per call): reference = frame 3 of the previous batch; it has detections -> becomes the reference
    o._interact(_ticket([20.]), [True])
    assert _refs(m) == [13.] and float(o.pre_dict["feat"][0, 0, 0, 0]) == 20.
    # a frame without detections: no interaction at all, the reference does not advance
    n = len(m.calls)
    assert o._interact(_ticket([30.]), [False]) is None and len(m.calls) == n and float(o.pre_dict["feat"][0, 0, 0, 0]) == 20.
    o._interact(_ticket([40.]), [True])
    assert _refs(m) == [20.]


def test_all_frames_with_detections_pair_t_with_t_minus_1():
    m = _FakeModel()
    o = OmniMOTFrame(m, None, (32, 32))
    o._interact(_ticket([1., 2., 3.]), [True, True, True])
    assert _refs(m) == [1., 1., 2.]                   # frame 1 of a sequence is its own reference (:1014-1015)
    o._interact(_ticket([4., 5.]), [True, True])
    assert _refs(m) == [3., 4.]


def test_reset_starts_a_new_video():
    """one object across videos must not pair video 2's first frame with video 1's last features
    (mot_evaluator.py:1014 re-seeds pre_dict at frame_id == 1, :978-979 builds a new tracker)."""
    import pytest
    m = _FakeModel()
    o = OmniMOTFrame(m, "tracker-1", (32, 32))
    o._interact(_ticket([1., 2.]), [True, True])
    o.frame_id = 2
    o.reset(tracker="tracker-2")
    assert o.pre_dict is None and o.frame_id == 0 and o.tracker == "tracker-2"
    o._interact(_ticket([7., 8.]), [False, True])
    assert _refs(m)[1] == 8.                            # own reference again, not 2.
    o._streaming = True                                # a run_stream generator is alive
    with pytest.raises(RuntimeError):
        o.reset()
