"""SOT per-frame driver on the HIP path; mirrors external/lib/test/tracker/unicorn_sot.py (UnicornSOTTrack).

Same control flow and thresholds (confthre 0.001, nmsthre 0.65, max_inst 3, index-0 box, int truncation);
the tensor logic of get_det_results (:78-109) runs on unicorn_amd kernels: the dense HWxHW correlation + softmax +
propagation is one fused kernel (uni_corr_softmax_pv) instead of a materialised 16000x16000 fp16 matrix.
Images are taken already letter-boxed as (1,3,H,W) BGR 0-255 float tensors or raw HxWx3 uint8 RGB arrays.
"""
import numpy as np
import torch

from ..ops import corr_softmax_pv, label_map_s8, letterbox, prior_pyramid
from ..ops import postprocess_launch
from ..utils.boxes import postprocess
from ..utils.timing import NoTimer


class UnicornSOTTrack:
    def __init__(self, model, input_size=(800, 1280), device="cuda"):
        self.model = model
        self.input_size = tuple(input_size)
        self.num_classes = 1
        self.confthre = 0.001        # unicorn_sot.py:23
        self.nmsthre = 0.65
        self.max_inst = 3
        self.device = device
        self.state = None
        self.frame_id = 0
        self.t = NoTimer()           # bench.py sets a StageTimer here for the per-stage numbers

    def _prep(self, image):
        """PreprocessorX.process (unicorn_sot.py:111-123): RGB->BGR, cv2-style 8-bit bilinear resize by r=min(H/h,W/w), pad 114."""
        if torch.is_tensor(image) and image.dim() == 4:
            return image.to(self.device).float(), 1.0
        return letterbox(np.asarray(image), self.input_size, swap_rb=True, device=self.device)   # uni_letterbox (post.hip)

    def initialize(self, image, info):
        self.frame_id = 0
        ref, r = self._prep(image)
        box = torch.tensor(info["init_bbox"], dtype=torch.float32).view(-1).clone()
        box[2:] += box[:2]
        box = box * r
        with torch.no_grad():
            _, self.out_dict_pre = self.model(imgs=ref, mode="backbone")
        self.dh, self.dw = self.out_dict_pre["h"] * 2, self.out_dict_pre["w"] * 2
        self.lbs_pre = label_map_s8(box, self.input_size[0], self.input_size[1], self.device)
        self.state = list(info["init_bbox"])

    def _network(self, cur):
        """unicorn_sot.py:78-108 up to the raw head outputs (no host sync)"""
        with torch.no_grad():
            fpn, d_cur = self.model(imgs=cur, mode="backbone")
            self.t.mark("backbone+fpn")
            f_pre, f_cur = self.model(seq_dict0=self.out_dict_pre, seq_dict1=d_cur, mode="interaction")
            e_pre = self.model(feat=f_pre, mode="upsample")
            e_cur = self.model(feat=f_cur, mode="upsample")
            self.t.mark("interaction+upsample")
            pred = corr_softmax_pv(e_pre.flatten(-2).squeeze(0), e_cur.flatten(-2).squeeze(0), self.lbs_pre,
                                   precision=0 if getattr(self.model, "precision", "f16x2") == "fp32" else 2)
            coarse = pred.view(1, -1, self.dh, self.dw)
            self.t.mark("correlation")
            outputs = self.model.head(fpn, prior_pyramid(coarse), mode="sot")
            outputs = outputs[0] if isinstance(outputs, tuple) else outputs
            self.t.mark("head")
            return outputs

    def get_det_results(self, cur):
        outputs = self._network(cur)
        det = postprocess(outputs, self.num_classes, self.confthre, self.nmsthre)[0]
        self.t.mark("postprocess")
        return det

    # ---- pipelined form of track(): submit enqueues a frame (H2D, letterbox, network, uni_postprocess, async read-back of the count
    # and of the first max_inst rows into pinned memory), collect waits for THAT frame's event only.  track() = collect(submit()).
    def submit(self, image):
        cur, r = self._prep(image)
        self.t.mark("h2d+letterbox")
        outputs = self._network(cur)
        post = postprocess_launch(outputs[0], self.num_classes, self.confthre, self.nmsthre)
        rows = torch.empty((self.max_inst, 7), dtype=torch.float32).pin_memory()
        rows.copy_(post.det[:self.max_inst], non_blocking=True)      # rows past the survivor count are never read
        ev = torch.cuda.Event()
        ev.record()
        self.t.mark("postprocess")
        return (post, rows, ev, r)

    def collect(self, ticket):
        post, rows, ev, r = ticket
        self.frame_id += 1
        ev.synchronize()
        m = int(post.n_host[0])
        if m > 0:
            output = rows[:min(m, self.max_inst)].numpy().copy()
            output[:, 0:4:2] = np.clip(output[:, 0:4:2], 0, self.input_size[1])      # unicorn_sot.py:64-65 (same fp32 values as the device clamp)
            output[:, 1:4:2] = np.clip(output[:, 1:4:2], 0, self.input_size[0])
            b = output[:, 0:4] / r
            b[:, 2] -= b[:, 0]
            b[:, 3] -= b[:, 1]
            self.state = [int(v) for v in b[0]]
        self.t.mark("box")
        return {"target_bbox": self.state}

    def track(self, image, info=None):
        return self.collect(self.submit(image))

    def track_stream(self, images):
        """images: iterable of frames of ONE sequence -> yields track()'s result per frame with one frame of lookahead: frame t+1 is
        enqueued before the host blocks on frame t's read-back (the SOT step of a frame depends only on the cached first frame,
        unicorn_sot.py:78-108), so the GPU never idles behind the host.  Same boxes as per-frame track() calls."""
        prev = None
        for img in images:
            tk = self.submit(img)
            if prev is not None:
                yield self.collect(prev)
            prev = tk
        if prev is not None:
            yield self.collect(prev)
