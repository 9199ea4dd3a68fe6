"""CPU restatement of the mask post-processing of the VOS / MOTS drivers (row N1 of SURVEY.md §8f).

TEST INFRASTRUCTURE ONLY (see oracle/unicorn_oracle.py).  Integer / byte work, so the HIP kernels are held to these
functions BIT-EXACTLY; the float part (bilinear resize) is written with an explicit fp32 operation order that the kernel
repeats without FMA contraction, and is itself held to torch's F.interpolate (<= 1 ulp) in tests/test_mask_oracle_cpu.py.

  resize_bilinear        F.interpolate(mask, scale_factor=1/r, mode="bilinear", align_corners=False)[:, 0, :H, :W]
                         (external/lib/test/tracker/unicorn_vos.py:146-150, unicorn/evaluators/mot_evaluator.py:804-805)
  soft_aggregate         unicorn_vos.py:99-120 (background = prod(1 - p), argmax over [background, ids], id map)
  overlap_free           mot_evaluator.py:860-865 (earlier tracks win overlapping pixels)
  rle_encode / rle_string / rle_decode
                         pycocotools.mask.encode on a Fortran-ordered mask + the "counts" string (mot_evaluator.py:889-892).
                         pycocotools is third-party and absent offline: restated from its published maskApi.c (rleEncode,
                         rleToString, rleFrString) -> "parity unpinned" for the byte format itself; pinned by the
                         round trip decode(encode(m)) == m through the independent rleFrString restatement and by the
                         hand-checked strings in the tests.
"""
import numpy as np

f32 = np.float32


def resize_src_index(n_out: int, n_in: int, rscale: np.float32):
    """area_pixel_compute_source_index (align_corners=False) + guard_index_and_lambda of ATen UpSample.h, in fp32"""
    d = np.arange(n_out, dtype=np.float32)
    real = (rscale * (d + f32(0.5))).astype(np.float32) - f32(0.5)
    real = np.maximum(real, f32(0.0)).astype(np.float32)
    i0 = np.minimum(np.floor(real).astype(np.int64), n_in - 1)
    lam1 = np.minimum(np.maximum(real - i0.astype(np.float32), f32(0.0)), f32(1.0)).astype(np.float32)
    i1 = i0 + (i0 < n_in - 1)
    return i0, i1, (f32(1.0) - lam1).astype(np.float32), lam1


def resize_bilinear(m: np.ndarray, r: float, H: int, W: int) -> np.ndarray:
    """(N, Hn, Wn) float32 -> (N, H, W) float32: bilinear by scale_factor 1/r, cropped to (H, W), zero outside the resized
    area (the drivers paste it into a zero (H, W) map).  out = wy0 * (wx0 * a + wx1 * b) + wy1 * (wx0 * c + wx1 * d), each
    product / sum rounded to fp32 in this order."""
    m = np.asarray(m, dtype=np.float32)
    N, Hn, Wn = m.shape
    sf = 1.0 / r
    ho, wo = int(np.floor(Hn * sf)), int(np.floor(Wn * sf))
    rscale = f32(1.0 / sf)
    y0, y1, wy0, wy1 = resize_src_index(ho, Hn, rscale)
    x0, x1, wx0, wx1 = resize_src_index(wo, Wn, rscale)
    a, b = m[:, y0][:, :, x0], m[:, y0][:, :, x1]
    c, d = m[:, y1][:, :, x0], m[:, y1][:, :, x1]
    top = ((wx0[None, None] * a).astype(np.float32) + (wx1[None, None] * b).astype(np.float32)).astype(np.float32)
    bot = ((wx0[None, None] * c).astype(np.float32) + (wx1[None, None] * d).astype(np.float32)).astype(np.float32)
    up = ((wy0[None, :, None] * top).astype(np.float32) + (wy1[None, :, None] * bot).astype(np.float32)).astype(np.float32)
    out = np.zeros((N, H, W), dtype=np.float32)
    hh, ww = min(H, ho), min(W, wo)
    out[:, :hh, :ww] = up[:, :hh, :ww]
    return out


def soft_aggregate(probs: np.ndarray, prob_ids, init_masks=None, init_ids=()):
    """unicorn_vos.py:99-120.  probs (K1, H, W) float32 probabilities of the tracked objects (list order = cur_obj_ids order),
    init_masks (K2, H, W) {0,1} masks of objects introduced in this frame.  Background = prod over the list of (1 - p) in
    fp32; np.argmax over channels [0 = background, id = its map] (first maximum wins: background, then the lowest id)."""
    probs = np.asarray(probs, dtype=np.float32)
    K1, H, W = probs.shape
    ids = [int(k) for k in prob_ids] + [int(k) for k in init_ids]
    nch = (max(ids) + 1) if ids else 1
    merge = np.zeros((H, W, nch), dtype=np.float32)
    bg = np.ones((H, W), dtype=np.float32)
    for k in range(K1):
        merge[:, :, int(prob_ids[k])] = probs[k]
        bg = (bg * (f32(1.0) - probs[k])).astype(np.float32)
    for k in range(len(init_ids)):
        mk = (np.asarray(init_masks[k]) != 0).astype(np.float32)
        merge[:, :, int(init_ids[k])] = mk
        bg = (bg * (f32(1.0) - mk)).astype(np.float32)
    merge[:, :, 0] = bg
    lab = np.argmax(merge, axis=-1)
    final = np.zeros((H, W), dtype=np.uint8)
    for k in ids:
        final[lab == k] = k
    return final


def overlap_free(masks: np.ndarray) -> np.ndarray:
    """mot_evaluator.py:860-865: masks (N, H, W) {0,1}, in track order; a pixel stays with the FIRST mask that claims it"""
    masks = np.asarray(masks) != 0
    out = masks.copy()
    prev = np.zeros(masks.shape[1:], dtype=bool)
    for n in range(masks.shape[0]):
        out[n] = masks[n] & ~prev
        prev |= masks[n]
    return out.astype(np.uint8)


def rle_encode(mask: np.ndarray) -> np.ndarray:
    """maskApi.c rleEncode: run lengths of the COLUMN-major (Fortran) pixel sequence, starting with the run of zeros"""
    t = (np.asarray(mask) != 0).astype(np.uint8).flatten(order="F")
    a = t.size
    if a == 0:
        return np.zeros((1,), dtype=np.uint32)
    change = np.flatnonzero(np.diff(np.concatenate(([0], t))) != 0)        # positions j with t[j] != t[j-1], t[-1] := 0
    bounds = np.concatenate((change, [a]))
    cnts = np.diff(np.concatenate(([0], bounds)))
    return cnts.astype(np.uint32)


def rle_string(cnts: np.ndarray) -> bytes:
    """maskApi.c rleToString: counts delta-coded against cnts[i-2] for i > 2, 5 data bits + continuation bit per char, +48"""
    out = bytearray()
    c = [int(v) for v in cnts]
    for i, v in enumerate(c):
        x = v - (c[i - 2] if i > 2 else 0)
        more = True
        while more:
            ch = x & 0x1f
            x >>= 5
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(ch + 48)
    return bytes(out)


def rle_from_string(s: bytes) -> np.ndarray:
    """maskApi.c rleFrString (the decoder pycocotools uses): independent restatement for the round-trip pin"""
    cnts = []
    p = 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = s[p] - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(cnts) > 2:
            x += cnts[-2]
        cnts.append(x)
    return np.asarray(cnts, dtype=np.int64)


def rle_decode(cnts, h: int, w: int) -> np.ndarray:
    v = np.zeros(h * w, dtype=np.uint8)
    pos, val = 0, 0
    for c in cnts:
        v[pos:pos + int(c)] = val
        pos += int(c)
        val ^= 1
    return v.reshape((h, w), order="F")


def mask_to_rle_string(mask: np.ndarray) -> bytes:
    """np.asfortranarray(mask) -> rletools.encode(mask)["counts"] (mot_evaluator.py:889-892)"""
    return rle_string(rle_encode(mask))
