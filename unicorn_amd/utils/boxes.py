"""Post-processing with the reference's signatures (unicorn/utils/boxes.py:33-152).

Row N1 of SURVEY.md §8f: `postprocess` / `postprocess_inst` run on the device through `uni_postprocess` (post.hip):
corners in place, class max + confidence filter, torchvision nms / batched_nms semantics (greedy, descending score, suppress
IoU > thr; batched = per class via the coordinate-offset trick), survivors in descending-score order.  One int32 is read
back per image (the row count).  `nms` / `batched_nms` below are the stand-alone torchvision-style entry points (used by
callers that bring their own boxes); they share the same kernels.
"""
import torch

from ..ops import condinst_masks, postprocess_image


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
    return [_select(ip, num_classes, conf_thre, nms_thre, class_agnostic)[0] for ip in prediction]


def postprocess_inst(prediction, locations, dynamic_params, fpn_levels, mask_feats, mask_head, num_classes, conf_thre=0.7,
                     nms_thre=0.45, class_agnostic=False, d_rate=4, up_masks=None, max_inst=None):
    """max_inst (extension): keep only the first max_inst detections before the masks are generated.  The reference computes a
    full-resolution mask for EVERY surviving detection and its drivers then read the first max_inst (unicorn_vos.py:134-136);
    slicing first gives the same rows / masks at 1/N of the work (4 MB per instance at 800x1280)."""
    output, output_mask = [], []
    for i, ip in enumerate(prediction):
        det, idx = _select(ip, num_classes, conf_thre, nms_thre, class_agnostic)
        if det is None:
            output.append(None)
            output_mask.append(None)
            continue
        if max_inst is not None:
            det, idx = det[:max_inst], idx[:max_inst]
        um = up_masks[0:1] if (up_masks is not None and len(up_masks) == 1) else (None if up_masks is None else up_masks[i:i + 1])
        # fused DynamicMaskHead + aligned_bilinear(d_rate) (boxes.py:138-146) on the surviving anchors
        lv = fpn_levels[i]
        lv = lv[idx if lv.is_cuda else idx.cpu()]                  # the reference keeps fpn_levels on the CPU (unicorn_head_mask.py:519)
        mf = mask_feats[0:1] if len(mask_feats) == 1 else mask_feats[i:i + 1]     # object-batched head: one image, K prediction sets
        masks = condinst_masks(mf, um, dynamic_params[i][idx], locations[idx], lv, mask_head.up_rate, d_rate)
        output.append(det)
        output_mask.append(masks)
    return output, output_mask
