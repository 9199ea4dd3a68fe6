"""Per-frame bodies of the reference's MOT / MOTS evaluation loops on the HIP path.

`OmniMOTFrame.run`  = unicorn/evaluators/mot_evaluator.py:991-1057 (`MOTEvaluator.evaluate_omni`): mode="whole" -> postprocess ->
(only if the frame has detections, :1005) interaction(reference frame, current frame) -> ONE embedding upsample -> instance embeddings
at the box centres (:1024-1034) -> rescale to the original image -> QuasiDenseEmbedTracker.match -> valid ids in ascending order
(:1049-1055).  The reference frame is the last frame THAT HAD DETECTIONS (`pre_dict = copy.deepcopy(cur_dict)` sits inside the
`if outputs[0] is not None` branch, :1005-1020).  DEVIATION, on purpose: the reference seeds `pre_dict` only `if frame_id == 1`
(:1014-1015) -- when frame 1 of a video has no detections it interacts the first frame that has some with the LAST VIDEO's features
(or raises NameError in the first video).  Here the first frame WITH detections of a video is its own reference, whatever its number.
One object serves ONE video: call `reset()` between videos (the reference re-seeds per video through frame_id == 1).
`OmniMOTSFrame.run` = :770-892 (the MOTS twin): postprocess_inst + CondInst masks, `> mask_thres` at the original resolution,
match(return_index=True), masks reordered to ascending track id, overlap-free merge (:860-865), pycocotools RLE strings (:889-892).

The evaluator classes themselves (data loader, result files, TrackEval glue) are out of scope; these two callables
are what their loop bodies do per frame, with every tensor step on unicorn_amd kernels.

Every call is cut into stages so that NO stage needs a stream-wide host sync (all read-backs go through pinned buffers + their own
events):
    A  GPU   `whole` + uni_postprocess launch (survivor count -> pinned)
    B  GPU   (host knows the counts) CondInst masks / thresholds, interaction + upsample, instance embeddings, async D2H of the rows
    H  host  native association (uni_qd_match), result rows; MOTS: enqueues C
    C  GPU   MOTS only: overlap-free merge + RLE strings (-> pinned)
`run` / `run_batch` execute A B H C back to back (one frame or B consecutive frames per call).  `run_stream(frames)` software-pipelines
consecutive calls on the ONE launch stream in the order  A0 A1 B0 A2 | H0 C0 B1 A3 | H1 C1 B2 A4 ...: the host association of frame t
and every read-back wait run while the GPU works on `whole` of a later frame, instead of the GPU idling behind the host as in the
reference loop (frame t+1's `whole` does not depend on frame t's association).  Results are identical to per-frame calls
(tests/test_model_gpu.py::test_omni_stream_pipelined_equals_per_frame).
"""
from collections import deque
from types import SimpleNamespace

import numpy as np
import torch

from ..ops import (condinst_masks, mots_overlap_free, nhwc, postprocess_collect, postprocess_launch, rle_encode_collect,
                   rle_encode_launch, sample_embeddings)
from ..utils.masks import mots_condinst_threshold, mots_threshold
from ..utils.timing import NoTimer


class _PinnedPool:
    """read-back buffers of the per-frame loops: pinned (rows, cols) fp32 buffers are taken from / returned to a small free list
    instead of one `pin_memory()` allocation per frame and tensor (capacity = rows rounded up to a power of two, >= 64)"""

    def __init__(self, keep=16):
        self.free, self.keep = {}, keep

    def to_pinned(self, t):
        """async device -> pinned host copy on the current stream (the caller records / waits an event) -> (view, base)"""
        n, cols = int(t.shape[0]), int(t.shape[1])
        cap = 64
        while cap < n:
            cap *= 2
        lst = self.free.setdefault((cap, cols, t.dtype), [])
        base = lst.pop() if lst else torch.empty((cap, cols), dtype=t.dtype).pin_memory()
        view = base[:n]
        view.copy_(t, non_blocking=True)
        return view, base

    def release(self, base):
        lst = self.free.setdefault((int(base.shape[0]), int(base.shape[1]), base.dtype), [])
        if len(lst) < self.keep:
            lst.append(base)


class OmniMOTFrame:
    def __init__(self, model, tracker, img_size, num_classes=1, confthre=0.01, nmsthre=0.7, embed_score_thr=0.1, timer=None):
        self.model, self.tracker = model, tracker
        self.img_size = tuple(img_size)
        self.num_classes, self.confthre, self.nmsthre = num_classes, confthre, nmsthre    # tools/track.py:100-101,156-159
        self.embed_score_thr = embed_score_thr                                              # mot_evaluator.py:1010
        self.pre_dict = None         # seq_dict of the last frame that had detections (:1020)
        self.frame_id = 0
        self._streaming = False      # a run_stream generator of this object is alive (frames in flight on the launch stream)
        self._pin = _PinnedPool()
        self.t = timer or NoTimer()

    def reset(self, tracker=None):
        """start a new video: forget the reference frame and the frame counter (mot_evaluator.py:1014 re-seeds at frame_id == 1);
        `tracker` optionally replaces the association state (the reference builds a new tracker per video, :968-975)"""
        if self._streaming:
            raise RuntimeError("reset() while a run_stream pipeline of this object is in flight: exhaust or close the generator first")
        self.pre_dict, self.frame_id = None, 0
        if tracker is not None:
            self.tracker = tracker

    # ---- stage A: network + postprocess launch over a batch of consecutive frames (no host sync)
    def _stage_a(self, imgs, info_img):
        m = self.model
        outputs, cur = m(imgs, mode="whole")                                                # :991
        self.t.mark("whole")
        out = outputs[0] if isinstance(outputs, tuple) else outputs
        post = [postprocess_launch(out[b], self.num_classes, self.confthre, self.nmsthre) for b in range(imgs.shape[0])]   # :995
        self.t.mark("postprocess")
        return SimpleNamespace(B=imgs.shape[0], info=info_img, outputs=outputs, cur=cur, post=post)

    # ---- reference frames of a batch (host knows which frames have detections)
    def _interact(self, tk, has_det):
        """interaction + upsample for the frames with detections -> embed (B,C,H8,W8) | None; advances self.pre_dict (:1014-1020)"""
        m, cur = self.model, tk.cur
        feat = cur["feat"]
        if not any(has_det):
            return None
        refs, last = [], (None if self.pre_dict is None else self.pre_dict["feat"])
        for b in range(tk.B):
            if has_det[b]:
                refs.append(feat[b:b + 1] if last is None else last)      # first frame with detections: its own reference (:1014-1015)
                last = feat[b:b + 1]
            else:
                refs.append(feat[b:b + 1])                                  # no detections: the reference skips the frame; its row is unused
        pre = dict(cur)
        pre["feat"] = nhwc(torch.cat(refs, 0)) if tk.B > 1 else refs[0]
        _, new_feat_cur = m(seq_dict0=pre, seq_dict1=cur, mode="interaction")               # :1017
        embed = m(feat=new_feat_cur, mode="upsample")                                       # :1019
        self.pre_dict = {"feat": last.clone(), "pos": cur["pos"][0:1], "h": cur["h"], "w": cur["w"]}    # :1020 (deepcopy)
        self.t.mark("interaction+upsample")
        return embed

    # ---- stage B: everything on the GPU that needs the survivor counts; ends with async D2H copies + one event
    def _stage_b(self, tk):
        dets = [postprocess_collect(p)[0] for p in tk.post]                                 # waits for the counts only
        embed = self._interact(tk, [d is not None for d in dets])
        img_h, img_w = tk.info
        scale = min(self.img_size[0] / float(img_h), self.img_size[1] / float(img_w))       # :1036-1038
        tk.rows = []
        for b, det in enumerate(dets):
            if det is None:
                tk.rows.append(None)
                continue
            bboxes, scores = det[:, :4], det[:, 4:5] * det[:, 5:6]                          # :1007
            # instance embeddings for EVERY detection (rows are independent); the score filter of :1009-1011 is applied on the host
            # copy, so no data-dependent shape is needed on the device
            feats = sample_embeddings(embed[b:b + 1], bboxes.contiguous())                  # :1024-1034 (uni_sample_embeddings)
            tk.rows.append(self._pin.to_pinned(torch.cat((bboxes / scale, scores), dim=1)) + self._pin.to_pinned(feats))   # :1039-1042
        self.t.mark("embeddings")
        tk.ev_b = torch.cuda.Event()
        tk.ev_b.record()

    # ---- host: association of every frame of the ticket, in order
    def _match_frame(self, ti, tf):
        keep = ti[:, 4] > self.embed_score_thr                                              # :1009-1011
        ti, tf = ti[keep], tf[keep]
        labels = torch.ones((ti.shape[0],))
        return keep, self.tracker.match(ti, labels, tf, self.frame_id)                      # :1045

    def _host_assoc(self, tk):
        tk.ev_b.synchronize()
        self.t.mark("d2h")
        tk.res = []
        for rows in tk.rows:
            self.frame_id += 1
            if rows is None:
                tk.res.append((None, None))
                continue
            ti, ti_base, tf, tf_base = rows
            _, (out_b, _, out_ids) = self._match_frame(ti, tf)
            self._pin.release(ti_base)               # match() copied what it keeps (numpy copies inside the native call wrapper)
            self._pin.release(tf_base)
            valid = out_ids > -1                                                            # :1047-1051
            out_b, out_ids = out_b[valid], out_ids[valid]
            _, inds = out_ids.sort(descending=False)                                        # :1052-1055
            tk.res.append((out_b[inds], out_ids[inds]))
        self.t.mark("association")

    def _finish(self, tk):
        return tk.res

    # ---- public entry points
    def run_batch(self, imgs, info_img):
        """imgs (B,3,H,W) consecutive frames -> list of B (bboxes (M,5), ids (M,)) | (None, None)"""
        tk = self._stage_a(imgs, info_img)
        self._stage_b(tk)
        self._host_assoc(tk)
        return self._finish(tk)

    def run(self, imgs, info_img):
        return self.run_batch(imgs, info_img)[0]

    def run_stream(self, frames, info_img):
        """frames: iterable of (B,3,H,W) batches (B = 1: one frame per call) of ONE video in order -> yields run_batch's result per
        item, software-pipelined over the launch stream (module docstring)."""
        if self._streaming:
            raise RuntimeError("run_stream: another pipeline of this object is still in flight")
        it = iter(frames)
        pend = deque()

        def admit():
            img = next(it, None)
            if img is not None:
                pend.append(self._stage_a(img, info_img))
        self._streaming = True
        try:
            admit()
            admit()
            cur = None
            if pend:
                cur = pend.popleft()
                self._stage_b(cur)
            admit()
            while cur is not None:
                self._host_assoc(cur)                    # waits for B(cur) only; the GPU keeps working on the admitted frames
                nxt = None
                if pend:
                    nxt = pend.popleft()
                    self._stage_b(nxt)
                admit()
                yield self._finish(cur)
                cur = nxt
        finally:
            self._streaming = False


class OmniMOTSFrame(OmniMOTFrame):
    def __init__(self, model, tracker, img_size, num_classes=1, confthre=0.01, nmsthre=0.7, embed_score_thr=0.1, mask_thres=0.3,
                 d_rate=2, min_box_area=100, timer=None):
        super().__init__(model, tracker, img_size, num_classes, confthre, nmsthre, embed_score_thr, timer)
        self.mask_thres, self.d_rate, self.min_box_area = mask_thres, d_rate, min_box_area  # :805, exp.d_rate, args.min_box_area
        self.fused_masks = True      # False: the two-pass path (network-size fp32 masks of the reference API, then uni_mask_resize): A/B + tests
        self._levels_dev = {}

    def _stage_b(self, tk):
        m = self.model
        outputs, locations, dyn, levels, mask_feats, up_masks = tk.outputs
        sel = [postprocess_collect(p) for p in tk.post]                                     # (det, anchor idx) | (None, None)   :776-778
        key = (levels.shape, str(dyn.device))
        if key not in self._levels_dev:          # fpn_levels live on the CPU (unicorn_head_mask.py:519): one device copy per shape
            self._levels_dev[key] = levels.to(dyn.device)
        lv_dev = self._levels_dev[key]
        img_h, img_w = tk.info
        scale = min(self.img_size[0] / float(img_h), self.img_size[1] / float(img_w))
        tk.rows, tk.masks = [], []
        # CondInst masks of every detection (postprocess_inst, boxes.py:138-146) and their thresholded full-resolution maps (:804-805)
        for b, (det, idx) in enumerate(sel):
            if det is None:
                tk.masks.append(None)
                continue
            if self.fused_masks:      # CondInst -> 1/scale resize -> `> thr` bytes in one pass (uni_condinst_masks_u8): same bits, no fp32 maps
                tk.masks.append(mots_condinst_threshold(mask_feats[b:b + 1], up_masks[b:b + 1], dyn[b][idx], locations[idx], lv_dev[b][idx],
                                                        m.head.mask_head.up_rate, self.d_rate, scale, int(img_h), int(img_w), self.mask_thres))
                continue
            om = condinst_masks(mask_feats[b:b + 1], up_masks[b:b + 1], dyn[b][idx], locations[idx], lv_dev[b][idx], m.head.mask_head.up_rate,
                                self.d_rate)
            tk.masks.append(mots_threshold(om, scale, int(img_h), int(img_w), self.mask_thres))
        self.t.mark("postprocess+condinst")
        embed = self._interact(tk, [d is not None for d, _ in sel])
        for b, (det, idx) in enumerate(sel):
            if det is None:
                tk.rows.append(None)
                continue
            bboxes, scores = det[:, :4], det[:, 4:5] * det[:, 5:6]
            feats = sample_embeddings(embed[b:b + 1], bboxes.contiguous())
            tk.rows.append(self._pin.to_pinned(torch.cat((bboxes / scale, scores), dim=1)) + self._pin.to_pinned(feats))
        self.t.mark("masks+embeddings")
        tk.ev_b = torch.cuda.Event()
        tk.ev_b.record()

    def _host_assoc(self, tk):
        tk.ev_b.synchronize()
        self.t.mark("d2h")
        tk.pending = []
        for rows, masks in zip(tk.rows, tk.masks):
            self.frame_id += 1
            if rows is None:
                tk.pending.append(None)
                continue
            ti, ti_base, tf, tf_base = rows
            keep = ti[:, 4] > self.embed_score_thr
            kept = torch.nonzero(keep)[:, 0]
            labels = torch.ones((int(kept.shape[0]),))
            out_b, _, out_ids, indexs = self.tracker.match(ti[keep], labels, tf[keep], self.frame_id, return_index=True)    # :843
            self._pin.release(ti_base)               # ti[keep] / tf[keep] are copies
            self._pin.release(tf_base)
            out_ids = torch.as_tensor(out_ids)
            valid = out_ids > -1
            idx = torch.nonzero(torch.as_tensor(indexs))[:, 0][valid]                       # masks[indexs][valid_inds] (:850-852)
            out_b, out_ids = torch.as_tensor(out_b)[valid], out_ids[valid]
            _, inds = out_ids.sort(descending=False)                                        # :853-856
            out_ids, out_b = out_ids[inds], out_b[inds]
            self.t.mark("association")
            order = kept[idx[inds]]                                                         # rows of the un-filtered detection list
            rle = None
            if order.numel():
                od = order.pin_memory().to(masks.device, non_blocking=True)
                free = mots_overlap_free(masks[od])                                         # :860-865 on the device
                rle = rle_encode_launch(free)                                               # :889-892 on the device, strings -> pinned
            tk.pending.append((out_b, out_ids, rle))
            self.t.mark("overlap-free+rle")

    def _finish(self, tk):
        res = []
        for p in tk.pending:
            if p is None:
                res.append(([], []))
                continue
            out_b, out_ids, rle = p
            rles = [b.decode("utf-8") for b in rle_encode_collect(rle)] if rle is not None else []
            ids, keep_rle = [], []
            for i in range(out_b.shape[0]):
                x1, y1, x2, y2 = [float(v) for v in out_b[i, :4]]
                if (x2 - x1) * (y2 - y1) > self.min_box_area:                               # :885
                    ids.append(int(out_ids[i]) + 1)
                    keep_rle.append(rles[i])
            res.append((ids, keep_rle))
        return res

    def run_batch(self, imgs, info_img):
        """-> list of B (online_ids (1-based), rle strings) like the rows appended to `results` (:893)"""
        tk = self._stage_a(imgs, info_img)
        self._stage_b(tk)
        self._host_assoc(tk)
        return self._finish(tk)

    def run_stream(self, frames, info_img):
        """as OmniMOTFrame.run_stream, with the C stage (overlap-free + RLE) of frame t enqueued right after its association and
        collected after B of the next frame and A of a later frame are in the queue"""
        return super().run_stream(frames, info_img)


class ByteMOTFrame(OmniMOTFrame):
    """Loop body of `MOTEvaluator.evaluate` (unicorn/evaluators/mot_evaluator.py:198-222) = what `tools/track.py` runs per frame:
    `outputs, _ = model(imgs)` (mode="whole", :199) -> `postprocess(outputs, num_classes, confthre, nmsthre)` (:203) ->
    `tracker.update(outputs[0], info_imgs, img_size)` (:212, the native BYTETracker) -> tracks with `tlwh[2] * tlwh[3] > min_box_area` that are
    not `vertical` (w / h > 1.6) (:216-222).  No embeddings, no interaction: stage B is only the survivor rows' copy to the host.
    `run(imgs, info_imgs)` -> (online_tlwhs, online_ids, online_scores) or None for a frame without detections (:211: the tracker is not
    stepped then); `run_stream` pipelines `whole` of later frames over the host association exactly like the `evaluate_omni` loop."""

    def __init__(self, model, tracker, img_size, num_classes=1, confthre=0.01, nmsthre=0.7, min_box_area=100, timer=None):
        super().__init__(model, tracker, img_size, num_classes, confthre, nmsthre, 0.0, timer)
        self.min_box_area = min_box_area                                                    # tools/track.py:110

    def _stage_b(self, tk):
        tk.rows = []
        for p in tk.post:
            det = postprocess_collect(p)[0]
            tk.rows.append(None if det is None else self._pin.to_pinned(det))               # (N, 7) [x1 y1 x2 y2 obj cls_conf cls]  boxes.py:71
        self.t.mark("rows")
        tk.ev_b = torch.cuda.Event()
        tk.ev_b.record()

    def _host_assoc(self, tk):
        tk.ev_b.synchronize()
        self.t.mark("d2h")
        tk.res = []
        for rows in tk.rows:
            self.frame_id += 1
            if rows is None:
                tk.res.append(None)
                continue
            det, base = rows
            targets = self.tracker.update(det.numpy(), tk.info, self.img_size)              # :212
            self._pin.release(base)
            tlwhs, ids, scores = [], [], []
            for t in targets:                                                               # :216-222
                tlwh = t.tlwh
                vertical = tlwh[2] / tlwh[3] > 1.6
                if tlwh[2] * tlwh[3] > self.min_box_area and not vertical:
                    tlwhs.append(tlwh)
                    ids.append(t.track_id)
                    scores.append(t.score)
            tk.res.append((tlwhs, ids, scores))
        self.t.mark("association")


class DemoPredictor:
    """`Predictor.inference` of tools/demo.py:136-172 for the detection / tracking experiments (`exp.task != "inst"`): letterbox
    (`ValTransform` = preproc, data_augment.py:194-214, no mean / std) -> `outputs, seq_dict = model(img)` (mode="whole", :166; the file as shipped
    has a dangling `try / except` there) -> `postprocess(outputs, num_classes, confthre, nmsthre, class_agnostic=True)`
    (:169-172).  `inference(img)` takes the HWC uint8 BGR image (cv2.imread's layout) -> (outputs list, img_info) like the reference; the
    letterbox runs on the device (uni_letterbox)."""

    def __init__(self, model, num_classes, confthre, nmsthre, test_size):
        self.model, self.num_classes, self.confthre, self.nmsthre, self.test_size = model, num_classes, confthre, nmsthre, tuple(test_size)

    def inference(self, img):
        from ..ops import letterbox
        from ..utils.boxes import postprocess
        height, width = img.shape[:2]
        ratio = min(self.test_size[0] / img.shape[0], self.test_size[1] / img.shape[1])      # :150
        img_info = {"id": 0, "file_name": None, "height": height, "width": width, "raw_img": img, "ratio": ratio}
        x, _ = letterbox(torch.as_tensor(img).cuda(), self.test_size, swap_rb=False)         # ValTransform: BGR kept, pad 114, CHW fp32 0-255
        with torch.no_grad():
            outputs, _ = self.model(x)                                                       # :166
            out = outputs[0] if isinstance(outputs, tuple) else outputs
            dets = postprocess(out, self.num_classes, self.confthre, self.nmsthre, class_agnostic=True)      # :169-172
        return dets, img_info
