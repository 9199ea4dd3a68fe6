"""oracle/mask_oracle.py (row N1 restatements) pinned on the CPU: the bilinear resize against torch's own F.interpolate, the
soft aggregation, the MOTS threshold and the overlap-free merge against outputs of the reference's OWN lines (exec'd by
tests/golden/make_golden_vos.py: unicorn_vos.py:99-120, mot_evaluator.py:804-805, 860-865), the RLE codec by hand-derived strings and the
encode -> decode round trip through the independent rleFrString restatement."""
import numpy as np
import torch
import torch.nn.functional as F

import mask_oracle as mo
import unicorn_oracle as uo


def test_resize_matches_torch_interpolate():
    g = torch.Generator().manual_seed(0)
    for (Hn, Wn, r, H, W) in [(80, 128, 0.5, 160, 256), (96, 160, 0.8333333, 110, 190), (64, 64, 1.0, 64, 64), (100, 160, 1.37, 70, 100)]:
        m = torch.rand(3, Hn, Wn, generator=g)
        ref = F.interpolate(m[:, None], scale_factor=1 / r, mode="bilinear", align_corners=False)[:, 0, :H, :W]
        got = mo.resize_bilinear(m.numpy(), r, H, W)
        assert got.shape == (3, H, W)
        hh, ww = ref.shape[1:]
        assert np.abs(got[:, :hh, :ww] - ref.numpy()).max() < 2e-7
        assert (got[:, hh:] == 0).all() and (got[:, :, ww:] == 0).all()


def _gold():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vos_mots_ref.npz"))


def test_soft_aggregate_matches_reference_lines():
    """external/lib/test/tracker/unicorn_vos.py:99-120 EXEC'D from the reference tree (tests/golden/make_golden_vos.py): tracked
    objects, objects of later reference groups, objects introduced in this frame, exact ties, certain rows -- against both
    restatements (mask_oracle.soft_aggregate = what the HIP merge kernel is held to, unicorn_oracle.vos_merge = what the model
    tests use)."""
    g = _gold()
    for tag in "abcd":
        probs, ids = g["vos_%s_probs" % tag], [str(k) for k in g["vos_%s_ids" % tag]]
        init, init_ids = g["vos_%s_init" % tag], [str(k) for k in g["vos_%s_init_ids" % tag]]
        ref = g["vos_%s_final" % tag]
        got = mo.soft_aggregate(probs, ids, init, init_ids)
        assert got.dtype == ref.dtype and np.array_equal(got, ref), tag
        d = {k: probs[i] for i, k in enumerate(ids)}
        for j, k in enumerate(init_ids):
            d[k] = init[j] != 0
        assert np.array_equal(uo.vos_merge(d, *ref.shape), ref), tag


def test_mots_threshold_and_overlap_free_match_reference_lines():
    """unicorn/evaluators/mot_evaluator.py:804-805 (resize by 1/scale, crop, > mask_thres) and :860-865 (overlap-free masks)
    EXEC'D from the reference tree on smooth probability maps that cross the threshold."""
    g = _gold()
    for tag in "abcd":
        prob = g["mots_%s_prob" % tag]
        scale, img_h, img_w = g["mots_%s_geom" % tag]
        shape = tuple(int(v) for v in g["mots_%s_shape" % tag])
        n = int(np.prod(shape))
        ref = np.unpackbits(g["mots_%s_masks" % tag])[:n].reshape(shape)
        free = np.unpackbits(g["mots_%s_free" % tag])[:n].reshape(shape)
        up = mo.resize_bilinear(prob, float(scale), int(img_h), int(img_w))[:, :shape[1], :shape[2]]
        got = (up > np.float32(0.5)).astype(np.uint8)
        assert np.array_equal(got, ref), (tag, int((got != ref).sum()))
        assert np.array_equal(mo.overlap_free(ref), free), tag


def test_rle_hand_checked_and_round_trip():
    assert mo.rle_string(np.array([4])) == b"4"                          # 2x2 zeros
    assert mo.mask_to_rle_string(np.ones((2, 2))) == b"04"
    assert mo.rle_string(np.array([33])) == b"Q1"                        # 33 = 1 + 32: low group 1 with continuation (1|32+48='Q'), then 1
    assert mo.rle_string(np.array([5, 3, 7, 1])) == b"537N"              # 4th count is coded as 1 - 3 = -2 -> 0b11110 + 48 = 'N'
    m = np.zeros((3, 4), dtype=np.uint8)
    m[1, 0] = m[2, 0] = m[0, 1] = 1                                      # column-major sequence 0 1 1 | 1 0 0 | 0 ... -> runs 1, 3, 8
    assert mo.rle_encode(m).tolist() == [1, 3, 8]
    g = np.random.default_rng(2)
    for shape in [(1, 1), (7, 13), (37, 29), (120, 200)]:
        for dens in (0.0, 0.02, 0.5, 1.0):
            mk = (g.random(shape) < dens).astype(np.uint8)
            if shape == (120, 200):                                      # blobs: long runs, large deltas
                mk[30:90, 50:140] = 1
            s = mo.mask_to_rle_string(mk)
            cn = mo.rle_from_string(s)
            assert np.array_equal(cn, mo.rle_encode(mk).astype(np.int64))
            assert np.array_equal(mo.rle_decode(cn, *shape), mk)


def test_overlap_free():
    m = np.zeros((3, 2, 4), dtype=np.uint8)
    m[0, 0, :2] = 1
    m[1, 0, 1:3] = 1
    m[2, :, :] = 1
    o = mo.overlap_free(m)
    assert o[0].tolist() == [[1, 1, 0, 0], [0, 0, 0, 0]] and o[1].tolist() == [[0, 0, 1, 0], [0, 0, 0, 0]]
    assert o[2].tolist() == [[0, 0, 0, 1], [1, 1, 1, 1]] and (o.sum(0) <= 1).all()


def test_product_rle_reader_inverts_the_oracle_encoder():
    """unicorn_amd.utils.masks.rle_string_to_mask (host-side reader of gathered MOTS strings) is the inverse of the pycocotools string
    format as restated in oracle/mask_oracle.py (rleToString of the column-major runs)."""
    import numpy as np
    from unicorn_amd.utils.masks import rle_string_to_mask
    import mask_oracle as mo
    rng = np.random.RandomState(0)
    for h, w in [(7, 5), (135, 240), (64, 1)]:
        m = (rng.rand(h, w) > 0.6).astype(np.uint8)
        m[: h // 3] = 0
        s = mo.mask_to_rle_string(m)
        assert np.array_equal(rle_string_to_mask(s, h, w), m)
        assert np.array_equal(rle_string_to_mask(s.decode("utf-8"), h, w), m)      # mots_rle hands out utf-8 decoded str (mot_evaluator.py:891)
    import pytest
    with pytest.raises(ValueError):
        rle_string_to_mask(mo.mask_to_rle_string(np.ones((4, 4), np.uint8)), 5, 4)
