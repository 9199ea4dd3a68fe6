"""Post-processing with the reference's signatures (unicorn/utils/boxes.py:33-152).

Row N1 of SURVEY.md §8f: `postprocess` / `postprocess_inst` run on the device through `uni_postprocess` (post.hip):
corners in place, class max + confidence filter, torchvision nms / batched_nms semantics (greedy, descending score, suppress
IoU > thr; batched = per class via the coordinate-offset trick), survivors in descending-score order.  One int32 is read
back per image (the row count).  `nms` / `batched_nms` below are the stand-alone torchvision-style entry points (used by
callers that bring their own boxes); they share the same kernels.
"""
import torch

from ..ops import condinst_masks, postprocess_collect, postprocess_image, postprocess_launch


_nms_ws = {}


def nms(boxes, scores, thr):
    """torchvision.ops.nms on the device (uni_nms): kept indices in descending-score order."""
    from .. import _lib as L
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.long, device=boxes.device)
    b, sc = boxes.float().contiguous(), scores.float().contiguous()
    need = L.lib().uni_nms_workspace_bytes(n)
    key = (b.device.index, torch.cuda.current_stream().cuda_stream)
    ws = _nms_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, device=b.device, dtype=torch.uint8)
        _nms_ws[key] = ws
    keep = torch.empty((n,), device=b.device, dtype=torch.int32)
    cnt = torch.zeros((1,), device=b.device, dtype=torch.int32)
    L.check(L.lib().uni_nms(L.ptr(b), L.ptr(sc), n, float(thr), L.ptr(keep), L.ptr(cnt), L.ptr(ws), ws.numel(), L.stream_ptr()), "uni_nms")
    return keep[:int(cnt.item())].long()


def batched_nms(boxes, scores, idxs, thr):
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.long, device=boxes.device)
    off = idxs.to(boxes) * (boxes.max() + 1)
    return nms(boxes + off[:, None], scores, thr)


def _corners_(prediction):
    c = prediction.new_empty(prediction.shape[:-1] + (4,))
    c[..., 0] = prediction[..., 0] - prediction[..., 2] / 2
    c[..., 1] = prediction[..., 1] - prediction[..., 3] / 2
    c[..., 2] = prediction[..., 0] + prediction[..., 2] / 2
    c[..., 3] = prediction[..., 1] + prediction[..., 3] / 2
    prediction[..., :4] = c          # in place, like the reference (boxes.py:39)
    return prediction


def _select(image_pred, num_classes, conf_thre, nms_thre, class_agnostic):
    """one image: (det (M,7) | None, anchor indices (M,) | None); image_pred is converted to corners in place"""
    return postprocess_image(image_pred, num_classes, conf_thre, nms_thre, class_agnostic)


def postprocess(prediction, num_classes, conf_thre=0.7, nms_thre=0.45, class_agnostic=False):
    tickets = [postprocess_launch(ip, num_classes, conf_thre, nms_thre, class_agnostic) for ip in prediction]      # all launches, then the read-backs
    return [postprocess_collect(t)[0] for t in tickets]


_lv_cache = {}


def _levels_on(fpn_levels, dev):
    """the reference keeps fpn_levels on the CPU (unicorn_head_mask.py:519) and indexes them with `idx.cpu()` -- one stream drain per image.
    They depend on the shapes only (the head hands out the same cached tensor every call): one device copy per tensor object."""
    import weakref
    if dev is None or fpn_levels.is_cuda:
        return fpn_levels
    key = (fpn_levels.data_ptr(), tuple(fpn_levels.shape), str(dev))
    hit = _lv_cache.get(key)
    if hit is None or hit[0]() is not fpn_levels or hit[2] != fpn_levels._version:
        if len(_lv_cache) > 16:
            _lv_cache.clear()
        hit = (weakref.ref(fpn_levels), fpn_levels.to(dev), fpn_levels._version)
        _lv_cache[key] = hit
    return hit[1]


def postprocess_inst(prediction, locations, dynamic_params, fpn_levels, mask_feats, mask_head, num_classes, conf_thre=0.7,
                     nms_thre=0.45, class_agnostic=False, d_rate=4, up_masks=None, max_inst=None):
    """max_inst (extension): keep only the first max_inst detections before the masks are generated.  The reference computes a
    full-resolution mask for EVERY surviving detection and its drivers then read the first max_inst (unicorn_vos.py:134-136);
    slicing first gives the same rows / masks at 1/N of the work (4 MB per instance at 800x1280)."""
    # every image's / object's uni_postprocess is enqueued before the first survivor count is read back (one wait for the last launch
    # instead of one stream drain per row of `prediction`: 0.27 ms each in the VOS step, tools/vos_profile.py)
    tickets = [postprocess_launch(ip, num_classes, conf_thre, nms_thre, class_agnostic) for ip in prediction]
    sel = [postprocess_collect(t) for t in tickets]
    if max_inst is not None:
        sel = [(d, i) if d is None else (d[:max_inst], i[:max_inst]) for d, i in sel]
    n_img = len(prediction)
    shared = len(mask_feats) == 1 and (up_masks is None or len(up_masks) == 1)     # object-batched head: ONE image, K prediction sets
    output = [d for d, _ in sel]
    output_mask = [None] * n_img
    live = [i for i in range(n_img) if sel[i][0] is not None]

    lv_dev = _levels_on(fpn_levels, prediction[0].device if n_img else None)

    def lv_of(i, idx):
        return lv_dev[i][idx]
    if shared and len(live) > 1:
        # the K objects of a VOS group share mask_feats / up_masks: ONE fused DynamicMaskHead + aligned_bilinear call (boxes.py:138-146) over
        # the surviving anchors of all of them instead of K calls of three launches each; rows are independent, results identical
        cnt = [int(sel[i][1].shape[0]) for i in live]
        params = torch.cat([dynamic_params[i][sel[i][1]] for i in live], 0)
        locs = torch.cat([locations[sel[i][1]] for i in live], 0)
        lvs = torch.cat([lv_of(i, sel[i][1]) for i in live], 0)
        masks = condinst_masks(mask_feats[0:1], None if up_masks is None else up_masks[0:1], params, locs, lvs, mask_head.up_rate, d_rate)
        off = 0
        for i, c in zip(live, cnt):
            output_mask[i] = masks[off:off + c]
            off += c
    else:
        for i in live:
            idx = sel[i][1]
            um = up_masks[0:1] if (up_masks is not None and len(up_masks) == 1) else (None if up_masks is None else up_masks[i:i + 1])
            mf = mask_feats[0:1] if len(mask_feats) == 1 else mask_feats[i:i + 1]
            # fused DynamicMaskHead + aligned_bilinear(d_rate) (boxes.py:138-146) on the surviving anchors
            output_mask[i] = condinst_masks(mf, um, dynamic_params[i][idx], locations[idx], lv_of(i, idx), mask_head.up_rate, d_rate)
    return output, output_mask
