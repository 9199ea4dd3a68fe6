"""`lib.test.tracker.unicorn_vos.UnicornVOSTrack` on the HIP path (external/lib/test/tracker/unicorn_vos.py:13-200).

Row N3 of SURVEY.md §8f: the reference re-runs the head once per object; here all objects of a reference group share ONE
correlation call (K value rows) and ONE batched head call (K prior sets over the same FPN maps).  The per-object selection,
the resize back to the original resolution and the soft aggregation (unicorn_vos.py:105-155) are host glue on device tensors
exactly like the reference.
"""
import numpy as np
import torch
import torch.nn.functional as F

from ..ops import corr_softmax_pv, label_map_s8, letterbox, prior_pyramid
from ..utils.boxes import postprocess_inst


class UnicornVOSTrack:
    def __init__(self, model, input_size=(800, 1280), device="cuda", d_rate=2, object_batched=False):
        self.model = model
        self.input_size = tuple(input_size)
        self.device = device
        self.num_classes = 1
        self.confthre = 0.001          # unicorn_vos.py:24-27
        self.nmsthre = 0.65
        self.max_inst = 1
        self.d_rate = d_rate
        # True: one correlation + one batched head call for all objects (row N3); False: the reference's per-object loop over
        # the same kernels.  Measured with synthetic weights at 800x1280: the loop is faster up to ~8 objects (14 vs 26-35 ms at
        # K=3, 32 vs 34 ms at K=8) because the K-fold FPN broadcast and the B=K head run small, launch-bound GEMMs.
        self.object_batched = object_batched
        self.frame_id = 0

    def _prep(self, image):
        if torch.is_tensor(image) and image.dim() == 4:
            return image.to(self.device).float(), 1.0
        return letterbox(np.asarray(image), self.input_size, swap_rb=True, device=self.device)

    def _label(self, box_xywh, r):
        box = torch.tensor(box_xywh, dtype=torch.float32).view(-1).clone()
        box[2:] += box[:2]                                  # (x1, y1, x2, y2), unicorn_vos.py:62-64
        return label_map_s8(box * r, self.input_size[0], self.input_size[1], self.device)

    def initialize(self, image, info):
        """info: init_object_ids (list), init_bbox {obj_id: xywh} (unicorn_vos.py:43-68)"""
        self.frame_id = 0
        self.init_object_ids = list(info["init_object_ids"])
        self.H, self.W = (image.shape[-2:] if torch.is_tensor(image) and image.dim() == 4 else np.asarray(image).shape[:2])
        ref, r = self._prep(image)
        with torch.no_grad():
            _, self.out_dict_pre = self.model(imgs=ref, mode="backbone")
        self.dh, self.dw = self.out_dict_pre["h"] * 2, self.out_dict_pre["w"] * 2
        self.lbs_pre_dict = {k: self._label(info["init_bbox"][k], r) for k in self.init_object_ids}

    def get_det_results(self, fpn, d_cur, d_pre, object_ids):
        """unicorn_vos.py:157-200, object batched: {obj_id: det (N,7) | None}, {obj_id: masks (N,1,H,W) | None}"""
        with torch.no_grad():
            f_pre, f_cur = self.model(seq_dict0=d_pre, seq_dict1=d_cur, mode="interaction")
            e_pre = self.model(feat=f_pre, mode="upsample")
            e_cur = self.model(feat=f_cur, mode="upsample")
            values = torch.cat([self.lbs_pre_dict[k] for k in object_ids], 0)                 # (K, HW/64)
            prec = 0 if getattr(self.model, "precision", "bf16") == "fp32" else 1
            pred = corr_softmax_pv(e_pre.flatten(-2).squeeze(0), e_cur.flatten(-2).squeeze(0), values, precision=prec)
            K = len(object_ids)
            coarse = pred.view(K, 1, self.dh, self.dw)
            p8, p16, p32 = prior_pyramid(coarse.transpose(0, 1).contiguous())                 # (1,K,..) pyramid
            pri = tuple(p.transpose(0, 1).contiguous() for p in (p8, p16, p32))               # (K,1,..): K prior sets
            outputs, locations, dyn, levels, mask_feats, up_masks = self.model.head(fpn, pri, mode="sot")
            out, out_mask = postprocess_inst(outputs, locations, dyn, levels, mask_feats, self.model.head.mask_head, self.num_classes,
                                             self.confthre, self.nmsthre, class_agnostic=False, d_rate=self.d_rate, up_masks=up_masks,
                                             max_inst=self.max_inst)
        return dict(zip(object_ids, out)), dict(zip(object_ids, out_mask))

    def step(self, image):
        """network-resolution result of one frame: {obj_id: (best det row | None, mask (H_in, W_in) | None)}, r"""
        cur, r = self._prep(image)
        with torch.no_grad():
            fpn, d_cur = self.model(imgs=cur, mode="backbone")
        if self.object_batched:
            det, msk = self.get_det_results(fpn, d_cur, self.out_dict_pre, self.init_object_ids)
        else:
            det, msk = {}, {}
            for k in self.init_object_ids:
                d1, m1 = self.get_det_results(fpn, d_cur, self.out_dict_pre, [k])
                det.update(d1)
                msk.update(m1)
        res = {}
        for k in self.init_object_ids:
            if det[k] is None:
                res[k] = (None, None)
                continue
            d = det[k].clone()
            d[:, 0:4:2] = d[:, 0:4:2].clamp(min=0, max=self.input_size[1])
            d[:, 1:4:2] = d[:, 1:4:2].clamp(min=0, max=self.input_size[0])
            res[k] = (d[0], msk[k][0, 0])
        return res, r

    def track(self, image, info=None):
        """unicorn_vos.py:71-121 for the objects of the first frame: {"segmentation": (H, W) uint8}"""
        self.frame_id += 1
        res, r = self.step(image)
        prob = {}
        for k in self.init_object_ids:
            m = res[k][1]
            full = np.zeros((self.H, self.W), dtype=np.float32)
            if m is not None:                                                 # :141-150
                up = F.interpolate(m[None, None], scale_factor=1 / r, mode="bilinear", align_corners=False)[0, 0, :self.H, :self.W]
                full[:up.shape[0], :up.shape[1]] = up.cpu().numpy()
            prob[k] = full
        ids = [int(k) for k in self.init_object_ids]
        merge = np.zeros((self.H, self.W, max(ids) + 1))                      # soft aggregation (:105-121)
        for k in self.init_object_ids:
            merge[:, :, int(k)] = prob[k]
        merge[:, :, 0] = np.prod(1 - np.stack([prob[k] for k in self.init_object_ids], axis=-1), axis=-1)
        lab = np.argmax(merge, axis=-1)
        final = np.zeros((self.H, self.W), dtype=np.uint8)
        for k in ids:
            final[lab == k] = k
        return {"segmentation": final}
