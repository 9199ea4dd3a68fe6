"""`lib.test.tracker.unicorn_vos.UnicornVOSTrack` on the HIP path (external/lib/test/tracker/unicorn_vos.py:13-200).

Same control flow as the reference driver: objects of the first frame form the first reference group, objects that appear
mid-sequence (`info["init_object_ids"]` / `init_bbox` / `init_mask` passed to `track`, :87-98) get the frame they appear in as
their own reference (`out_dict_pre_new` / `obj_ids_new`), every frame runs `get_det_results` once per group and the per-object
mask probabilities are soft-aggregated into one id map (:99-120).

What differs is where the work happens (rows N3 / N1 of SURVEY.md §8f):
  * per group, interaction + the two embedding upsamples + the HW x HW correlation run ONCE, with the K label maps of the
    group as K value rows of one `uni_corr_softmax_pv` call (the reference recomputes nothing either, but materialises the
    16000^2 fp16 similarity);
  * the head runs ONCE per group over K prior sets (`uni_head_objects`: FPN casts, stem convs and the whole mask branch are
    shared, only prior fusion / attention blocks / towers / predictions / controllers run per object) instead of K full passes;
  * resize to the original resolution, background product, argmax and id map are one kernel (`uni_vos_merge`) on the device:
    only the final (H, W) uint8 map is copied back.
"""
import numpy as np
import torch

from ..ops import corr_softmax_pv, label_map_s8, letterbox, prior_pyramid, vos_merge
from ..utils.boxes import postprocess_inst


class UnicornVOSTrack:
    def __init__(self, model, input_size=(800, 1280), device="cuda", d_rate=2, object_batched=True, max_objects_per_call=16):
        self.model = model
        self.input_size = tuple(input_size)
        self.device = device
        self.num_classes = 1
        self.confthre = 0.001          # unicorn_vos.py:24-27
        self.nmsthre = 0.65
        self.max_inst = 1
        self.mask_thres = 0.30
        self.d_rate = d_rate
        # True (default): one head call over all objects of a group (uni_head_objects); False: the reference's per-object loop
        # over the same kernels (kept as a tested option; tools/vos_bench.py compares the two)
        self.object_batched = object_batched
        self.max_objects_per_call = max_objects_per_call
        self.frame_id = 0

    # ------------------------------------------------------------------------------------------------ helpers
    def _prep(self, image):
        if torch.is_tensor(image) and image.dim() == 4:
            return image.to(self.device).float(), 1.0
        return letterbox(np.asarray(image), self.input_size, swap_rb=True, device=self.device)

    def _label(self, box_xywh, r):
        box = torch.tensor(box_xywh, dtype=torch.float32).view(-1).clone()
        box[2:] += box[:2]                                  # (x1, y1, x2, y2), unicorn_vos.py:62-64
        return label_map_s8(box * r, self.input_size[0], self.input_size[1], self.device)

    # ------------------------------------------------------------------------------------------------ driver
    def initialize(self, image, info):
        """info: init_object_ids (list), init_bbox {obj_id: xywh on the original image} (unicorn_vos.py:43-69)"""
        self.frame_id = 0
        self.init_object_ids = list(info["init_object_ids"])
        self.sequence_object_ids = list(info.get("sequence_object_ids", self.init_object_ids))
        self.H, self.W = (image.shape[-2:] if torch.is_tensor(image) and image.dim() == 4 else np.asarray(image).shape[:2])
        ref, r = self._prep(image)
        with torch.no_grad():
            _, self.out_dict_pre = self.model(imgs=ref, mode="backbone")
        self.dh, self.dw = self.out_dict_pre["h"] * 2, self.out_dict_pre["w"] * 2
        self.lbs_pre_dict = {k: self._label(info["init_bbox"][k], r) for k in self.init_object_ids}
        self.state_pre_dict = {k: list(info["init_bbox"][k]) for k in self.init_object_ids}
        self.out_dict_pre_new, self.obj_ids_new = [], []    # reference frames of objects that appear later (:68-69)
        self._last_det = {}                                 # get_mask_results is callable on its own, like the reference method

    def get_det_results(self, fpn, d_cur, d_pre, object_ids):
        """unicorn_vos.py:157-200 for one reference group: {obj_id: det (N,7) | None}, {obj_id: masks (N,1,Hn,Wn) | None}"""
        m = self.model
        with torch.no_grad():
            f_pre, f_cur = m(seq_dict0=d_pre, seq_dict1=d_cur, mode="interaction")
            e_pre = m(feat=f_pre, mode="upsample")
            e_cur = m(feat=f_cur, mode="upsample")
            values = torch.cat([self.lbs_pre_dict[k] for k in object_ids], 0)                 # (K, HW/64): ONE correlation call
            prec = 0 if getattr(m, "precision", "f16x2") == "fp32" else 2       # fp32 MFMA in the exact mode, f16x2 split otherwise
            pred = corr_softmax_pv(e_pre.flatten(-2).squeeze(0), e_cur.flatten(-2).squeeze(0), values, precision=prec)
            K = len(object_ids)
            p8, p16, p32 = prior_pyramid(pred.view(1, K, self.dh, self.dw))                   # (1,K,..) pyramid of all objects
            pri = tuple(p.transpose(0, 1).contiguous() for p in (p8, p16, p32))               # (K,1,..): K prior sets
            out, out_mask = [], []
            step = self.max_objects_per_call if self.object_batched else 1
            for k0 in range(0, K, step):
                sl = slice(k0, min(K, k0 + step))
                outputs, locations, dyn, levels, mask_feats, up_masks = m.head(fpn, tuple(p[sl] for p in pri), mode="sot")
                o, om = postprocess_inst(outputs, locations, dyn, levels, mask_feats, m.head.mask_head, self.num_classes,
                                         self.confthre, self.nmsthre, class_agnostic=False, d_rate=self.d_rate, up_masks=up_masks,
                                         max_inst=self.max_inst)
                out += o
                out_mask += om
        return dict(zip(object_ids, out)), dict(zip(object_ids, out_mask))

    def get_mask_results(self, fpn, d_cur, d_pre, r, object_ids):
        """unicorn_vos.py:123-155: best instance per object -> {obj_id: network-resolution mask probabilities (Hn, Wn) on the
        device | None}, instance scores.  (The resize to the original resolution is fused into the aggregation kernel.)"""
        det, msk = self.get_det_results(fpn, d_cur, d_pre, object_ids)
        probs, scores = {}, np.zeros((len(object_ids),))
        # the best row of every object goes to the host in ONE copy (a `.cpu()` per object drains the stream K times per frame)
        live = [k for k in object_ids if det[k] is not None]
        rows = {}
        if live:
            top = torch.stack([det[k][0] for k in live])                                       # (n_live, 7): index-0 instance (:131-136)
            top[:, 0:4:2] = top[:, 0:4:2].clamp(min=0, max=self.input_size[1])
            top[:, 1:4:2] = top[:, 1:4:2].clamp(min=0, max=self.input_size[0])
            host = top.cpu().numpy()
            rows = {k: (top[i], host[i]) for i, k in enumerate(live)}
        for i, k in enumerate(object_ids):
            if k not in rows:
                probs[k] = None
                continue
            d0, dn = rows[k]
            b = dn[0:4] / r
            b[2] -= b[0]
            b[3] -= b[1]
            self.state_pre_dict[k] = [int(v) for v in b]                                       # :137-139
            scores[i] = dn[4] * dn[5]
            probs[k] = msk[k][0, 0]
            self._last_det[k] = d0
        return probs, scores

    def step(self, image):
        """network-resolution result of one frame for every tracked object (tests / benchmarks):
        {obj_id: (best det row | None, mask (Hn, Wn) | None)}, r"""
        cur, r = self._prep(image)
        with torch.no_grad():
            fpn, d_cur = self.model(imgs=cur, mode="backbone")
        self._last_det = {}
        res = {}
        for d_pre, ids in [(self.out_dict_pre, self.init_object_ids)] + list(zip(self.out_dict_pre_new, self.obj_ids_new)):
            probs, _ = self.get_mask_results(fpn, d_cur, d_pre, r, ids)
            for k in ids:
                res[k] = (self._last_det.get(k), probs[k])
        return res, r

    def track(self, image, info=None):
        """unicorn_vos.py:71-121 -> {"segmentation": (H, W) uint8 id map}"""
        info = info or {}
        self.frame_id += 1
        cur, r = self._prep(image)
        with torch.no_grad():
            fpn, d_cur = self.model(imgs=cur, mode="backbone")
        self._last_det = {}
        probs, order = {}, []
        # instances from the first frame, then from the intermediate frames (:79-85)
        for d_pre, ids in [(self.out_dict_pre, self.init_object_ids)] + list(zip(self.out_dict_pre_new, self.obj_ids_new)):
            p, _ = self.get_mask_results(fpn, d_cur, d_pre, r, ids)
            probs.update(p)
            order += list(ids)
        # instances that appear in the current frame (:87-98): this frame becomes their reference, their mask is the given one
        init_ids, init_masks = [], None
        if "init_object_ids" in info:
            init_ids = list(info["init_object_ids"])
            self.out_dict_pre_new.append(d_cur)
            self.obj_ids_new.append(init_ids)
            for k in init_ids:
                self.state_pre_dict[k] = list(info["init_bbox"][k])
                self.lbs_pre_dict[k] = self._label(info["init_bbox"][k], r)
            im = torch.as_tensor(np.asarray(info["init_mask"])).to(self.device)
            init_masks = torch.stack([(im == int(k)) for k in init_ids]).to(torch.uint8)
        # soft aggregation on the device (:99-120); objects without a detection contribute an all-zero map, i.e. nothing
        live = [k for k in order if probs[k] is not None]
        pt = torch.stack([probs[k] for k in live]) if live else None
        if pt is None and not init_ids:
            return {"segmentation": np.zeros((self.H, self.W), dtype=np.uint8)}
        seg = vos_merge(pt, live, r, self.H, self.W, init_masks, init_ids)
        return {"segmentation": seg.cpu().numpy()}
