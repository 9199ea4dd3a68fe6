"""Thin tensor-level wrappers over the C-ABI operators (torch only allocates and hands over pointers)."""
import ctypes as C

import torch

from . import _lib as L


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.UnicornHipError("unicorn_amd ops need HIP device tensors (got a CPU tensor); there is no CPU fallback")


def nhwc(t):
    """(1,C,H,W) tensor -> fp32 tensor whose memory is [H][W][C] (channels_last), no copy when already so."""
    t = t.float() if t.dtype != torch.float32 else t
    return t.contiguous(memory_format=torch.channels_last)


def empty_nhwc(c, h, w, device, b=1):
    return torch.empty((b, c, h, w), device=device, dtype=torch.float32, memory_format=torch.channels_last)


def msda_forward(value, spatial_shapes, level_start_index, sampling_locations, attention_weights, im2col_step=64):
    """Same signature as MultiScaleDeformableAttention.ms_deform_attn_forward (ops/src/vision.cpp:13-16)."""
    _need_cuda(value, sampling_locations, attention_weights)
    if value.dtype != torch.float32:
        raise L.UnicornHipError("msda_forward: only fp32 is implemented (reference dispatches fp32/fp64)")
    value, loc, attn = value.contiguous(), sampling_locations.contiguous(), attention_weights.contiguous()
    N, S, M, D = value.shape
    _, Lq, _, Ln, P, _ = loc.shape
    shp = (C.c_int64 * (2 * Ln))(*[int(v) for v in spatial_shapes.reshape(-1).tolist()])
    lsi = (C.c_int64 * Ln)(*[int(v) for v in level_start_index.reshape(-1).tolist()])
    out = torch.empty((N, Lq, M * D), device=value.device, dtype=torch.float32)
    if out.numel() == 0:
        return out
    L.check(L.lib().uni_msda_fwd(L.ptr(value), shp, lsi, L.ptr(loc), L.ptr(attn), L.ptr(out), N, S, M, D, Lq, Ln, P,
                                 L.stream_ptr()), "uni_msda_fwd")
    return out


_corr_ws = {}


def corr_softmax_pv(embed_ref, embed_cur, values, precision=0):
    """embed_* : (C, HW) (any strides; an NHWC map viewed as (C,HW) needs no copy) ; values (K, HW_ref) -> (K, HW_cur).
    == values @ softmax(embed_ref^T @ embed_cur, dim=0)   (unicorn_sot.py:95-100)
    precision 0: exact fp32 MFMA; 1: fp32-equivalent "bf16x3" (operands split exactly into three bf16 pieces, six fp32-exact
    partial products accumulated in fp32; error at the fp32 rounding level, 2.7x less MFMA time); 2: fp32-equivalent "f16x2"
    (hi + lo f16 halves, three products: half the MFMA time of bf16x3, same error class for embeddings inside the f16 range);
    3: the arithmetic class of the reference driver itself, which casts keys / queries / values to fp16 (:95-97): f16-rounded
    operands, one product, f16-rounded scores (the `.half()` of the normalised softmax is not reproduced).  The trackers keep 2."""
    _need_cuda(embed_ref, embed_cur, values)
    er = embed_ref.float().t().contiguous()     # (HW, C) row-major
    ec = embed_cur.float().t().contiguous()
    v = values.float().contiguous()
    R, D = er.shape
    Q = ec.shape[0]
    K = v.shape[0]
    out = torch.empty((K, Q), device=er.device, dtype=torch.float32)
    need = L.lib().uni_corr_workspace_bytes(R, Q, K)
    key = (er.device.index, torch.cuda.current_stream().cuda_stream)
    ws = _corr_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1), device=er.device, dtype=torch.uint8)
        _corr_ws[key] = ws
    L.check(L.lib().uni_corr_softmax_pv(L.ptr(er), L.ptr(ec), L.ptr(v), L.ptr(out), R, Q, D, K, precision, L.ptr(ws),
                                        ws.numel(), L.stream_ptr()), "uni_corr_softmax_pv")
    return out


def corr_softmax_pv_batched(embed_ref, embed_cur, values, precision=2, values_per_frame=False):
    """B frames in ONE launch (`uni_corr_softmax_pv_batched`): embed_* (B, C, HW) or (B, C, H, W) (channels_last maps need no copy),
    values (K, HW_ref) shared by the frames -- the SOT label map of the cached first frame -- or (B, K, HW_ref) with values_per_frame
    -> (B, K, HW_cur).  == torch.stack([corr_softmax_pv(embed_ref[b], embed_cur[b], values) for b in range(B)])."""
    _need_cuda(embed_ref, embed_cur, values)
    er = embed_ref.float().flatten(2).transpose(1, 2).contiguous()     # (B, HW, C)
    ec = embed_cur.float().flatten(2).transpose(1, 2).contiguous()
    v = values.float().contiguous()
    B, R, D = er.shape
    Q = ec.shape[1]
    K = v.shape[-2]
    if values_per_frame and (v.dim() != 3 or v.shape[0] != B):
        raise ValueError("values_per_frame needs values of shape (B, K, HW_ref)")
    if not values_per_frame and v.dim() != 2:
        raise ValueError("shared values must have shape (K, HW_ref)")
    if ec.shape[0] != B or v.shape[-1] != R:
        raise ValueError("corr_softmax_pv_batched: shape mismatch")
    out = torch.empty((B, K, Q), device=er.device, dtype=torch.float32)
    need = L.lib().uni_corr_workspace_bytes_batched(B, R, Q, K)
    key = (er.device.index, torch.cuda.current_stream().cuda_stream)
    ws = _corr_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1), device=er.device, dtype=torch.uint8)
        _corr_ws[key] = ws
    L.check(L.lib().uni_corr_softmax_pv_batched(L.ptr(er), L.ptr(ec), L.ptr(v), L.ptr(out), B, R, Q, D, K, 1 if values_per_frame else 0,
                                                precision, L.ptr(ws), ws.numel(), L.stream_ptr()), "uni_corr_softmax_pv_batched")
    return out


def prior_pyramid(coarse):
    """(1,K,H8,W8) -> (coarse, 1/2, 1/4) like unicorn_sot.py:103-105"""
    _need_cuda(coarse)
    c = coarse.float().contiguous()
    _, K, H8, W8 = c.shape
    p16 = torch.empty((1, K, H8 // 2, W8 // 2), device=c.device, dtype=torch.float32)
    p32 = torch.empty((1, K, H8 // 4, W8 // 4), device=c.device, dtype=torch.float32)
    L.check(L.lib().uni_prior_pyramid(L.ptr(c), L.ptr(p16), L.ptr(p32), K, H8, W8, L.stream_ptr()), "uni_prior_pyramid")
    return c, p16, p32


def label_map_s8(box_xyxy, H, W, device):
    b = torch.as_tensor(box_xyxy, dtype=torch.float32).reshape(4).to(device)
    out = torch.empty((1, (H // 8) * (W // 8)), device=device, dtype=torch.float32)
    L.check(L.lib().uni_label_map_s8(L.ptr(b), L.ptr(out), H, W, L.stream_ptr()), "uni_label_map_s8")
    return out


def sample_embeddings(embed, boxes_xyxy, stride=8.0):
    """embed (1,C,H8,W8); boxes (N,>=4) xyxy in input pixels -> (N,C)  (mot_evaluator.py:1024-1034)"""
    _need_cuda(embed, boxes_xyxy)
    e = nhwc(embed)
    _, Cc, H8, W8 = e.shape
    b = boxes_xyxy.float().contiguous()
    n = b.shape[0]
    out = torch.empty((n, Cc), device=e.device, dtype=torch.float32)
    if n:
        L.check(L.lib().uni_sample_embeddings(L.ptr(e), H8, W8, Cc, L.ptr(b), b.shape[1], n, float(stride), L.ptr(out),
                                              L.stream_ptr()), "uni_sample_embeddings")
    return out


def condinst_masks(mask_feats, up_masks, params, inst_loc, inst_lvl, up_rate, d_rate):
    """-> (N,1,d_rate*up_rate*H8, d_rate*up_rate*W8) sigmoid scores"""
    _need_cuda(mask_feats, up_masks, params)
    mf, um = nhwc(mask_feats), nhwc(up_masks)
    _, _, H8, W8 = mf.shape
    p = params.float().contiguous()
    n = p.shape[0]
    loc = inst_loc.float().contiguous().to(mf.device)
    lvl = inst_lvl.to(device=mf.device, dtype=torch.int32).contiguous()
    out = torch.empty((n, 1, d_rate * up_rate * H8, d_rate * up_rate * W8), device=mf.device, dtype=torch.float32)
    if n:
        ws = torch.empty(n * H8 * W8 * (1 + up_rate * up_rate), device=mf.device, dtype=torch.float32)
        L.check(L.lib().uni_condinst_masks(L.ptr(mf), L.ptr(um), L.ptr(p), p.shape[1], L.ptr(loc), L.ptr(lvl), n, H8, W8,
                                           up_rate, d_rate, L.ptr(out), L.ptr(ws), ws.numel() * 4, L.stream_ptr()),
                "uni_condinst_masks")
    return out


def condinst_masks_resized(mask_feats, up_masks, params, inst_loc, inst_lvl, up_rate, d_rate, r, H, W, thr=None):
    """condinst_masks + mask_resize in ONE call (uni_condinst_masks_u8): the CondInst scores of `params` resized by 1/r and pasted into
    (N, H, W) maps -- `> thr` bytes (mot_evaluator.py:804-805) or, with thr=None, fp32 probabilities (unicorn_vos.py:141-152) -- without the
    (N, 1, Hn, Wn) network-size maps ever reaching HBM.  Bit-identical to mask_resize(condinst_masks(...)[:, 0], r, H, W, thr)."""
    _need_cuda(mask_feats, up_masks, params)
    mf, um = nhwc(mask_feats), nhwc(up_masks)
    _, _, H8, W8 = mf.shape
    p = params.float().contiguous()
    n = p.shape[0]
    out = torch.empty((n, int(H), int(W)), device=mf.device, dtype=torch.float32 if thr is None else torch.uint8)
    if n:
        loc = inst_loc.float().contiguous().to(mf.device)
        lvl = inst_lvl.to(device=mf.device, dtype=torch.int32).contiguous()
        ws = torch.empty(n * H8 * W8 * (1 + up_rate * up_rate), device=mf.device, dtype=torch.float32)
        L.check(L.lib().uni_condinst_masks_u8(L.ptr(mf), L.ptr(um), L.ptr(p), p.shape[1], L.ptr(loc), L.ptr(lvl), n, H8, W8, up_rate, d_rate,
                                              float(r), int(H), int(W), 0.0 if thr is None else float(thr), L.ptr(out) if thr is None else None,
                                              None if thr is None else L.ptr(out), L.ptr(ws), ws.numel() * 4, L.stream_ptr()),
                "uni_condinst_masks_u8")
    return out


_post_ws = {}


class PostTicket:
    """uni_postprocess in flight: device buffers + the row count on its way to pinned host memory + the event that says it arrived"""
    __slots__ = ("det", "keep", "n_dev", "n_host", "event", "A")


def postprocess_launch(image_pred, num_classes, conf_thre, nms_thre, class_agnostic=False, precornered=False):
    """Enqueue unicorn/utils/boxes.py:33-77 for ONE image (uni_postprocess) WITHOUT a host sync: the survivor count goes to pinned
    host memory by an async copy and an event is recorded behind it.  `postprocess_collect(ticket)` waits for THAT event only, so
    work enqueued later on the same stream (the next frame of a pipelined tracker loop) does not delay the read-back."""
    _need_cuda(image_pred)
    if image_pred.dtype != torch.float32 or image_pred.stride(-1) != 1:
        raise L.UnicornHipError("postprocess_image: needs a float32 (A, 5+nc) tensor with unit inner stride")
    A, ld = image_pred.shape[0], image_pred.stride(0) if image_pred.shape[0] > 1 else image_pred.shape[1]
    dev = image_pred.device
    need = L.lib().uni_postprocess_workspace_bytes(A)
    key = (dev.index, torch.cuda.current_stream().cuda_stream)
    ws = _post_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1), device=dev, dtype=torch.uint8)
        _post_ws[key] = ws
    t = PostTicket()
    t.A = A
    t.det = torch.empty((max(A, 1), 7), device=dev, dtype=torch.float32)
    t.keep = torch.empty((max(A, 1),), device=dev, dtype=torch.int32)
    t.n_dev = torch.zeros((1,), device=dev, dtype=torch.int32)
    L.check(L.lib().uni_postprocess(L.ptr(image_pred), A, ld, num_classes, float(conf_thre), float(nms_thre),
                                    int(bool(class_agnostic)) | (2 if precornered else 0), A, L.ptr(t.det), L.ptr(t.keep), L.ptr(t.n_dev), L.ptr(ws), ws.numel(), L.stream_ptr()), "uni_postprocess")
    t.n_host = torch.empty((1,), dtype=torch.int32).pin_memory()
    t.n_host.copy_(t.n_dev, non_blocking=True)
    t.event = torch.cuda.Event()
    t.event.record()
    return t


def postprocess_collect(t):
    """-> (det (M,7), anchor indices (M,) int64) or (None, None); blocks on the ticket's own event only"""
    t.event.synchronize()
    m = int(t.n_host[0])
    if m == 0:
        return None, None
    return t.det[:m], t.keep[:m].long()


def postprocess_image(image_pred, num_classes, conf_thre, nms_thre, class_agnostic=False, precornered=False):
    """unicorn/utils/boxes.py:33-77 for ONE image on the device (uni_postprocess): image_pred (A, 5+nc) decoded cxcywh fp32,
    converted to corners in place (precornered=True: boxes are already xyxy).  Returns (det (M,7), anchor indices (M,) int64) or (None, None).  The only host sync is
    the read-back of M (the output shape is data dependent)."""
    return postprocess_collect(postprocess_launch(image_pred, num_classes, conf_thre, nms_thre, class_agnostic, precornered))


def letterbox(image, input_size, swap_rb=True, device=None):
    """PreprocessorX.process (unicorn_sot.py:111-123, swap_rb=True) / preproc (data_augment.py:194-214, swap_rb=False) on the
    device: image (h, w, 3) uint8 (numpy array or cuda tensor) -> ((1, 3, H, W) float32 cuda tensor, r)."""
    import ctypes as C
    if not torch.is_tensor(image):
        import numpy as np
        image = torch.from_numpy(np.ascontiguousarray(image))
    if image.dtype != torch.uint8 or image.dim() != 3 or image.shape[2] != 3:
        raise L.UnicornHipError("letterbox: needs an (h, w, 3) uint8 image")
    if not image.is_cuda:
        image = image.to(device if device is not None else "cuda", non_blocking=True)
    image = image.contiguous()
    H, W = int(input_size[0]), int(input_size[1])
    out = torch.empty((1, 3, H, W), device=image.device, dtype=torch.float32)
    r = C.c_double(0.0)
    with torch.cuda.device(image.device):
        L.check(L.lib().uni_letterbox(L.ptr(image), image.shape[0], image.shape[1], int(bool(swap_rb)), H, W, L.ptr(out), C.byref(r),
                                      L.stream_ptr()), "uni_letterbox")
    return out, r.value


# ---------------------------------------------------------------------------------------------------------------------
# mask post-processing of the VOS / MOTS drivers on the device (csrc/mask_post.hip)
# ---------------------------------------------------------------------------------------------------------------------
def mask_resize(masks, r, H, W, thr=None):
    """masks (N, Hn, Wn) fp32 cuda -> F.interpolate(masks[:, None], scale_factor=1/r, bilinear)[:, 0, :H, :W] pasted into zero
    (N, H, W) maps (unicorn_vos.py:146-150); with thr: the `> thr` byte masks of mot_evaluator.py:804-805 instead."""
    _need_cuda(masks)
    m = masks.float().contiguous()
    N, Hn, Wn = m.shape
    if N == 0:
        return torch.empty((0, H, W), device=m.device, dtype=torch.float32 if thr is None else torch.uint8)
    if thr is None:
        out = torch.empty((N, H, W), device=m.device, dtype=torch.float32)
        L.check(L.lib().uni_mask_resize(L.ptr(m), N, Hn, Wn, float(r), H, W, 0.0, L.ptr(out), None, L.stream_ptr()), "uni_mask_resize")
    else:
        out = torch.empty((N, H, W), device=m.device, dtype=torch.uint8)
        L.check(L.lib().uni_mask_resize(L.ptr(m), N, Hn, Wn, float(r), H, W, float(thr), None, L.ptr(out), L.stream_ptr()), "uni_mask_resize")
    return out


def vos_merge(probs, prob_ids, r, H, W, init_masks=None, init_ids=()):
    """soft aggregation of unicorn_vos.py:99-120 fused with the 1/r resize: probs (K1, Hn, Wn) network-resolution mask
    probabilities of the tracked objects (ids prob_ids, cur_obj_ids order), init_masks (K2, H, W) of objects introduced in
    this frame -> (H, W) uint8 id map on the device."""
    dev = probs.device if probs is not None else init_masks.device
    K1 = 0 if probs is None else probs.shape[0]
    K2 = len(init_ids)
    out = torch.empty((H, W), device=dev, dtype=torch.uint8)
    p = probs.float().contiguous() if K1 else None
    pid = torch.tensor([int(k) for k in prob_ids], dtype=torch.int32, device=dev) if K1 else None
    im = init_masks.to(torch.uint8).contiguous() if K2 else None
    iid = torch.tensor([int(k) for k in init_ids], dtype=torch.int32, device=dev) if K2 else None
    Hn, Wn = (p.shape[1], p.shape[2]) if K1 else (0, 0)
    L.check(L.lib().uni_vos_merge(L.ptr(p), L.ptr(pid), K1, Hn, Wn, float(r), L.ptr(im), L.ptr(iid), K2, H, W, L.ptr(out),
                                  L.stream_ptr()), "uni_vos_merge")
    return out


def mots_overlap_free(masks):
    """mot_evaluator.py:860-865: (N, H, W) bool/uint8 masks in track order -> earlier tracks keep overlapping pixels"""
    _need_cuda(masks)
    m = masks.to(torch.uint8).contiguous()
    out = torch.empty_like(m)
    if m.shape[0]:
        L.check(L.lib().uni_mots_overlap_free(L.ptr(m), m.shape[0], m.shape[1], m.shape[2], L.ptr(out), L.stream_ptr()), "uni_mots_overlap_free")
    return out


def rle_encode_launch(masks, max_runs=1 << 14):
    """enqueue uni_rle_encode + async copies of the string lengths and characters into pinned host memory; -> ticket"""
    _need_cuda(masks)
    m = masks.to(torch.uint8).contiguous()
    N, H, W = m.shape
    if N == 0:
        return {"N": 0}
    max_chars = 6 * (max_runs + 1)
    ws = torch.empty(L.lib().uni_rle_workspace_bytes(N, H, W, max_runs), device=m.device, dtype=torch.uint8)
    chars = torch.empty((N, max_chars), device=m.device, dtype=torch.uint8)
    lens = torch.empty((N,), device=m.device, dtype=torch.int32)
    L.check(L.lib().uni_rle_encode(L.ptr(m), N, H, W, max_runs, max_chars, L.ptr(chars), L.ptr(lens), None, None, L.ptr(ws),
                                   ws.numel(), L.stream_ptr()), "uni_rle_encode")
    # the strings of real masks are short (a few hundred characters): the first `head` characters of every row travel with the
    # lengths; a longer string (or an overflow of max_runs, length < 0) is fetched / re-encoded by rle_encode_collect
    head = min(max_chars, 4096)
    lens_h = torch.empty((N,), dtype=torch.int32).pin_memory()
    chars_h = torch.empty((N, head), dtype=torch.uint8).pin_memory()
    lens_h.copy_(lens, non_blocking=True)
    chars_h.copy_(chars[:, :head], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return {"N": N, "m": m, "chars": chars, "lens_h": lens_h, "chars_h": chars_h, "head": head, "event": ev, "max_runs": max_runs}


def rle_encode_collect(t):
    if t["N"] == 0:
        return []
    t["event"].synchronize()
    ln = t["lens_h"].tolist()
    if min(ln) < 0:                          # a mask with more runs than expected: synchronous retry with larger bounds
        return rle_encode(t["m"], max_runs=t["max_runs"] * 4)
    if max(ln) > t["head"]:
        host = t["chars"][:, :max(ln)].cpu().numpy()
    else:
        host = t["chars_h"].numpy()
    return [host[i, :ln[i]].tobytes() for i in range(t["N"])]


def rle_encode(masks, max_runs=1 << 14):
    """pycocotools.mask.encode(np.asfortranarray(mask))["counts"] for every (H, W) mask of an (N, H, W) {0,1} cuda tensor
    (mot_evaluator.py:889-892) -> list of bytes objects.  Runs are found and the strings are written on the device; one
    read-back of lengths + chars."""
    _need_cuda(masks)
    m = masks.to(torch.uint8).contiguous()
    N, H, W = m.shape
    if N == 0:
        return []
    while True:
        max_chars = 6 * (max_runs + 1)
        ws = torch.empty(L.lib().uni_rle_workspace_bytes(N, H, W, max_runs), device=m.device, dtype=torch.uint8)
        chars = torch.empty((N, max_chars), device=m.device, dtype=torch.uint8)
        lens = torch.empty((N,), device=m.device, dtype=torch.int32)
        L.check(L.lib().uni_rle_encode(L.ptr(m), N, H, W, max_runs, max_chars, L.ptr(chars), L.ptr(lens), None, None, L.ptr(ws),
                                       ws.numel(), L.stream_ptr()), "uni_rle_encode")
        ln = lens.cpu().tolist()
        if min(ln) >= 0:
            break
        max_runs *= 4                      # a mask with more runs than expected: retry with larger bounds
    top = max(ln)
    host = chars[:, :top].cpu().numpy()
    return [host[i, :ln[i]].tobytes() for i in range(N)]
