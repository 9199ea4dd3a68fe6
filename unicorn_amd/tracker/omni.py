"""Per-frame bodies of the reference's MOT / MOTS evaluation loops on the HIP path.

`OmniMOTFrame.run`  = unicorn/evaluators/mot_evaluator.py:991-1045 (`MOTEvaluator.evaluate_omni`): mode="whole" -> postprocess ->
interaction(previous frame, current frame) -> ONE embedding upsample -> instance embeddings at the box centres (:1024-1034) ->
rescale to the original image -> QuasiDenseEmbedTracker.match -> valid ids.
`OmniMOTSFrame.run` = :770-890 (the MOTS twin): postprocess_inst + CondInst masks, `> mask_thres` at the original resolution,
match(return_index=True), masks reordered to ascending track id, overlap-free merge (:860-865), pycocotools RLE strings (:889-892).

The evaluator classes themselves (data loader, result files, TrackEval glue) are out of scope; these two callables
are what their loop bodies do per frame, with every tensor step on unicorn_amd kernels.  Time-batching: `run_batch` takes B
consecutive frames -- `whole`, the interaction (frame t against frame t - 1: the previous frame's seq_dict is a backbone output, so
the whole batch is known) and the upsample run once over the batch; detection filtering, embedding sampling and the association are
per frame, in order.
"""
import copy

import torch

from ..ops import nhwc, sample_embeddings
from ..utils.boxes import postprocess, postprocess_inst
from ..utils.masks import mots_rle, mots_threshold
from ..utils.timing import NoTimer


class OmniMOTFrame:
    def __init__(self, model, tracker, img_size, num_classes=1, confthre=0.01, nmsthre=0.7, embed_score_thr=0.1, timer=None):
        self.model, self.tracker = model, tracker
        self.img_size = tuple(img_size)
        self.num_classes, self.confthre, self.nmsthre = num_classes, confthre, nmsthre    # tools/track.py:100-101,156-159
        self.embed_score_thr = embed_score_thr                                              # mot_evaluator.py:1010
        self.pre_dict = None
        self.frame_id = 0
        self.t = timer or NoTimer()

    # ---- network part over a batch of consecutive frames
    def _network(self, imgs):
        m = self.model
        outputs, cur = m(imgs, mode="whole")                                                # :991
        self.t.mark("whole")
        B = imgs.shape[0]
        feat = cur["feat"]
        if self.pre_dict is None:                                                           # frame 1: its own reference (:1014-1015)
            prev_last = feat[0:1]
        else:
            prev_last = self.pre_dict["feat"]
        pre = dict(cur)
        pre["feat"] = nhwc(torch.cat([prev_last, feat[:-1]], 0)) if B > 1 else prev_last    # frame t interacts with frame t - 1
        _, new_feat_cur = m(seq_dict0=pre, seq_dict1=cur, mode="interaction")               # :1017
        embed = m(feat=new_feat_cur, mode="upsample")                                       # :1019
        self.pre_dict = {"feat": feat[B - 1:B].clone(), "pos": cur["pos"][0:1], "h": cur["h"], "w": cur["w"]}   # :1020 (deepcopy)
        self.t.mark("interaction+upsample")
        return outputs, embed

    def _associate(self, det, embed_b, info_img):
        """one frame: det (N,7) rows of postprocess | None -> (output_bboxes (M,5), output_ids (M,))"""
        self.frame_id += 1
        if det is None:
            return None, None
        bboxes, scores = det[:, :4], det[:, 4:5] * det[:, 5:6]                              # :1007
        keep = scores[:, 0] > self.embed_score_thr                                          # :1009-1011
        bboxes, scores = bboxes[keep], scores[keep]
        labels = torch.ones((bboxes.size(0),))
        feats = sample_embeddings(embed_b, bboxes.contiguous())                             # :1024-1034 (uni_sample_embeddings)
        img_h, img_w = info_img
        scale = min(self.img_size[0] / float(img_h), self.img_size[1] / float(img_w))       # :1036-1038
        track_inputs = torch.cat((bboxes / scale, scores), dim=1)
        self.t.mark("embeddings")
        ti, tf = track_inputs.cpu(), feats.cpu()                                            # :1041-1042
        self.t.mark("d2h")
        out_b, _, out_ids = self.tracker.match(ti, labels, tf, self.frame_id)               # :1045
        valid = out_ids > -1
        self.t.mark("association")
        return out_b[valid], out_ids[valid]

    def run_batch(self, imgs, info_img):
        """imgs (B,3,H,W) consecutive frames -> list of B (bboxes, ids)"""
        outputs, embed = self._network(imgs)
        outputs = outputs[0] if isinstance(outputs, tuple) else outputs
        dets = postprocess(outputs, self.num_classes, self.confthre, self.nmsthre)          # :995 (uni_postprocess per image)
        self.t.mark("postprocess")
        return [self._associate(dets[b], embed[b:b + 1], info_img) for b in range(imgs.shape[0])]

    def run(self, imgs, info_img):
        return self.run_batch(imgs, info_img)[0]


class OmniMOTSFrame(OmniMOTFrame):
    def __init__(self, model, tracker, img_size, num_classes=1, confthre=0.01, nmsthre=0.7, embed_score_thr=0.1, mask_thres=0.3,
                 d_rate=2, min_box_area=100, timer=None):
        super().__init__(model, tracker, img_size, num_classes, confthre, nmsthre, embed_score_thr, timer)
        self.mask_thres, self.d_rate, self.min_box_area = mask_thres, d_rate, min_box_area  # :804, exp.d_rate, args.min_box_area

    def run_batch(self, imgs, info_img):
        """-> list of B (online_ids (1-based), rle strings) like the rows appended to `results` (:889)"""
        m = self.model
        det_outputs, embed = self._network(imgs)
        outputs, locations, dyn, levels, mask_feats, up_masks = det_outputs
        img_h, img_w = info_img
        scale = min(self.img_size[0] / float(img_h), self.img_size[1] / float(img_w))
        res = []
        for b in range(imgs.shape[0]):
            o, om = postprocess_inst(outputs[b:b + 1], locations, dyn[b:b + 1], levels[b:b + 1], mask_feats[b:b + 1], m.head.mask_head,
                                     self.num_classes, self.confthre, self.nmsthre, class_agnostic=False, d_rate=self.d_rate,
                                     up_masks=up_masks[b:b + 1])                            # :776-778
            self.t.mark("postprocess+condinst")
            self.frame_id += 1
            det = o[0]
            if det is None:
                res.append(([], []))
                continue
            bboxes, scores = det[:, :4], det[:, 4:5] * det[:, 5:6]
            masks = mots_threshold(om[0], scale, int(img_h), int(img_w), self.mask_thres)   # :804-805 (uni_mask_resize)
            keep = scores[:, 0] > self.embed_score_thr
            bboxes, scores, masks = bboxes[keep], scores[keep], masks[keep]
            labels = torch.ones((bboxes.size(0),))
            feats = sample_embeddings(embed[b:b + 1], bboxes.contiguous())
            track_inputs = torch.cat((bboxes / scale, scores), dim=1)
            self.t.mark("masks+embeddings")
            ti, tf = track_inputs.cpu(), feats.cpu()
            self.t.mark("d2h")
            out_b, _, out_ids, indexs = self.tracker.match(ti, labels, tf, self.frame_id, return_index=True)    # :843
            out_ids = torch.as_tensor(out_ids)
            valid = out_ids > -1
            idx = torch.nonzero(torch.as_tensor(indexs))[:, 0][valid]                       # masks[indexs][valid_inds] (:850-851)
            out_b, out_ids = torch.as_tensor(out_b)[valid], out_ids[valid]
            _, inds = out_ids.sort(descending=False)                                        # :853-856
            out_ids, out_b = out_ids[inds], out_b[inds]
            self.t.mark("association")
            free, rles = mots_rle(masks, order=idx[inds].tolist())                          # :860-865 + :889-892 on the device
            ids, keep_rle = [], []
            for i in range(out_b.shape[0]):
                x1, y1, x2, y2 = [float(v) for v in out_b[i, :4]]
                if (x2 - x1) * (y2 - y1) > self.min_box_area:                               # :877
                    ids.append(int(out_ids[i]) + 1)
                    keep_rle.append(rles[i])
            self.t.mark("overlap-free+rle")
            res.append((ids, keep_rle))
        return res
