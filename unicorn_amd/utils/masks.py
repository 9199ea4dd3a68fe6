"""Mask post-processing of the MOTS evaluator (unicorn/evaluators/mot_evaluator.py:803-890) on the device (row N1).

    masks = F.interpolate(outputs_mask[0], scale_factor=1/scale, bilinear)[:, 0, :img_h, :img_w] > mask_thres     (:804-805)
    ... association picks / reorders the masks (indexs, valid_inds, ascending track id)                              (:843-856)
    overlap-free: masks_new[n] = masks[n] & ~(masks[0] | ... | masks[n-1])                                          (:860-865)
    rle = pycocotools.mask.encode(np.asfortranarray(mask))["counts"].decode("utf-8")                                (:889-892)

`mots_threshold` and `mots_rle` wrap the three kernels (uni_mask_resize, uni_mots_overlap_free, uni_rle_encode); only the RLE
strings (a few KB) leave the device.
"""
import torch

from ..ops import mask_resize, mots_overlap_free, rle_encode


def mots_threshold(outputs_mask, scale, img_h, img_w, mask_thres=0.30, crop=True):
    """outputs_mask (N, 1, Hn, Wn) sigmoid scores of postprocess_inst -> uint8 masks at the original resolution
    (mot_evaluator.py:804-805): F.interpolate(scale_factor=1/scale)[:, 0, :img_h, :img_w] > mask_thres.
    The interpolated map has floor(Hn / scale) x floor(Wn / scale) pixels, which can be ONE SHORT of the image (480 x 854 image at
    800 x 1280: 1280 / 1.4988 -> 853): the reference then encodes a (480, 853) mask.  crop=True (default) returns exactly that
    shape, so the RLE strings are the reference's; crop=False returns the zero-padded (N, img_h, img_w) maps the VOS driver
    uses (unicorn_vos.py:146-150 pastes into a full-size map)."""
    img_h, img_w = int(img_h), int(img_w)
    if outputs_mask is None or outputs_mask.shape[0] == 0:
        dev = outputs_mask.device if outputs_mask is not None else "cuda"
        return torch.zeros((0, img_h, img_w), dtype=torch.uint8, device=dev)
    out = mask_resize(outputs_mask[:, 0], scale, img_h, img_w, thr=mask_thres)
    if crop:
        import math
        ho = min(img_h, int(math.floor(outputs_mask.shape[2] * (1.0 / scale))))      # F.interpolate's output size: floor(in * scale_factor)
        wo = min(img_w, int(math.floor(outputs_mask.shape[3] * (1.0 / scale))))
        if ho < img_h or wo < img_w:
            out = out[:, :ho, :wo].contiguous()
    return out


def mots_condinst_threshold(mask_feats, up_masks, params, inst_loc, inst_lvl, up_rate, d_rate, scale, img_h, img_w, mask_thres=0.30, crop=True):
    """postprocess_inst's masks (utils/boxes.py:138-146) + mots_threshold in one fused call: the arguments of ops.condinst_masks for the kept
    detections of ONE image -> the uint8 masks mots_threshold(condinst_masks(...), scale, img_h, img_w, mask_thres, crop) returns, bit for
    bit, without the (N, 1, Hn, Wn) fp32 maps (uni_condinst_masks_u8)."""
    from ..ops import condinst_masks_resized
    img_h, img_w = int(img_h), int(img_w)
    out = condinst_masks_resized(mask_feats, up_masks, params, inst_loc, inst_lvl, up_rate, d_rate, scale, img_h, img_w, thr=mask_thres)
    if crop and out.shape[0]:
        import math
        Hn, Wn = d_rate * up_rate * mask_feats.shape[2], d_rate * up_rate * mask_feats.shape[3]
        ho, wo = min(img_h, int(math.floor(Hn * (1.0 / scale)))), min(img_w, int(math.floor(Wn * (1.0 / scale))))
        if ho < img_h or wo < img_w:
            out = out[:, :ho, :wo].contiguous()
    return out


def mots_rle(masks, order=None):
    """masks (N, H, W) uint8 in detection order, order: indices into masks in ascending-track-id order (after `indexs` /
    `valid_inds`, mot_evaluator.py:850-856).  Returns (overlap-free masks (M, H, W) uint8 on the device, list of M RLE strings)."""
    if order is not None:
        masks = masks[torch.as_tensor(order, dtype=torch.long, device=masks.device)]
    if masks.shape[0] == 0:
        return masks, []
    free = mots_overlap_free(masks)
    return free, [b.decode("utf-8") for b in rle_encode(free)]


def rle_string_to_mask(s, h, w):
    """Inverse of the strings `mots_rle` / `rle_encode` produce (pycocotools maskApi.c rleFrString + rleDecode): compressed counts string
    -> (h, w) uint8 numpy mask, column-major runs starting with zeros.  Host-side reader of gathered result strings."""
    import numpy as np
    if isinstance(s, str):                      # mots_rle returns utf-8 decoded strings (mot_evaluator.py:891), rle_encode raw bytes
        s = s.encode("ascii")
    cnts, p = [], 0
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
    if sum(cnts) != h * w:
        raise ValueError("RLE string does not describe a %d x %d mask (runs sum to %d)" % (h, w, sum(cnts)))
    v = np.zeros(h * w, dtype=np.uint8)
    pos, val = 0, 0
    for c in cnts:
        v[pos:pos + c] = val
        pos += c
        val ^= 1
    return v.reshape((h, w), order="F")
