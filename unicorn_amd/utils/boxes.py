"""Post-processing with the reference's signatures (unicorn/utils/boxes.py:33-152).

Row N1 of SURVEY.md §8f ("next"): host-side glue on device tensors.  NMS follows torchvision semantics (greedy,
descending score, suppress IoU > thr; batched = per-class via the coordinate-offset trick), evaluated as one
IoU matrix on the GPU plus a sequential sweep over the (few hundred) candidates.
"""
import torch

from ..ops import condinst_masks


def nms(boxes, scores, thr):
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.long, device=boxes.device)
    order = torch.argsort(scores, descending=True, stable=True)
    b = boxes[order]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(b[:, None, :2], b[None, :, :2])
    rb = torch.min(b[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    over = (inter / (area[:, None] + area[None] - inter) > thr).cpu()      # one D2H copy
    keep, sup = [], torch.zeros(n, dtype=torch.bool)
    for i in range(n):
        if sup[i]:
            continue
        keep.append(i)
        sup |= over[i]
    return order[torch.tensor(keep, dtype=torch.long, device=boxes.device)]


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
    class_conf, class_pred = torch.max(image_pred[:, 5:5 + num_classes], 1, keepdim=True)
    conf_mask = image_pred[:, 4] * class_conf.squeeze(1) >= conf_thre
    det = torch.cat((image_pred[:, :5], class_conf, class_pred.float()), 1)[conf_mask]
    if det.shape[0] == 0:
        return None, None, conf_mask
    sc = det[:, 4] * det[:, 5]
    keep = nms(det[:, :4], sc, nms_thre) if class_agnostic else batched_nms(det[:, :4], sc, det[:, 6], nms_thre)
    return det[keep], keep, conf_mask


def postprocess(prediction, num_classes, conf_thre=0.7, nms_thre=0.45, class_agnostic=False):
    prediction = _corners_(prediction)
    return [_select(ip, num_classes, conf_thre, nms_thre, class_agnostic)[0] for ip in prediction]


def postprocess_inst(prediction, locations, dynamic_params, fpn_levels, mask_feats, mask_head, num_classes, conf_thre=0.7,
                     nms_thre=0.45, class_agnostic=False, d_rate=4, up_masks=None):
    prediction = _corners_(prediction)
    output, output_mask = [], []
    for i, ip in enumerate(prediction):
        det, keep, conf_mask = _select(ip, num_classes, conf_thre, nms_thre, class_agnostic)
        if det is None:
            output.append(None)
            output_mask.append(None)
            continue
        cm_cpu = conf_mask.cpu()
        locs = locations[conf_mask][keep]
        dps = dynamic_params[i][conf_mask][keep]
        lvls = fpn_levels[i][cm_cpu][keep.cpu()]
        um = up_masks[0:1] if (up_masks is not None and len(up_masks) == 1) else (None if up_masks is None else up_masks[i:i + 1])
        # fused DynamicMaskHead + aligned_bilinear(d_rate) (boxes.py:138-146)
        masks = condinst_masks(mask_feats[i:i + 1], um, dps, locs, lvls, mask_head.up_rate, d_rate)
        output.append(det)
        output_mask.append(masks)
    return output, output_mask
