"""CPU restatement of the reference's input letterbox (row 0 / N1 of SURVEY.md §8):
  * PreprocessorX.process  external/lib/test/tracker/unicorn_sot.py:111-123 (RGB->BGR, cv2.resize INTER_LINEAR, pad 114, CHW float)
  * preproc                unicorn/data/data_augment.py:194-214            (same without the channel swap)

TEST INFRASTRUCTURE ONLY.  The arithmetic is OpenCV's (third-party, absent offline -> "parity unpinned"): cv2.resize with
INTER_LINEAR on 8-bit images is fixed point (modules/imgproc/src/resize.cpp, resizeGeneric_ + HResizeLinear<uchar,int,short> +
VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>): 11-bit coefficients `saturate_cast<short>(w * 2048)` (round half
to even) from `fx = (float)((dx + 0.5) * scale - 0.5)`, horizontal pass in int32, vertical pass
`(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2`.  Restated from that published source; the known-answer
checks in tests/test_oracle_golden.py (identity, constant images, exact 2x cases) hold it to properties any correct cv2 has.
"""
import numpy as np


def _coeffs(dst, src):
    """per destination index: (s0, s1, a0, a1) following resize.cpp (xofs / ialpha and yofs / ibeta)"""
    scale = float(src) / dst                                  # double
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)          # fx = (float)((dx+0.5)*scale_x - 0.5)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return s, f


def _rnd_short(x):
    """saturate_cast<short>(float) = cvRound (round half to even), saturated"""
    return np.clip(np.rint(x.astype(np.float32)), -32768, 32767).astype(np.int64)


def cv2_resize_linear_u8(img, dsize):
    """img (h, w, c) uint8 -> (dh, dw, c) uint8, dsize = (dw, dh) like cv2.resize"""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    h, w, _ = img.shape
    dw, dh = int(dsize[0]), int(dsize[1])
    sx, fx = _coeffs(dw, w)
    lo = sx < 0
    fx = np.where(lo, np.float32(0), fx); sx = np.where(lo, 0, sx)            # if (sx < 0) fx = 0, sx = 0
    hi = sx >= w - 1
    fx = np.where(hi, np.float32(0), fx); sx = np.where(hi, w - 1, sx)        # if (sx >= ssize.width-1) fx = 0, sx = width-1
    a0 = _rnd_short((np.float32(1.0) - fx) * np.float32(2048)); a1 = _rnd_short(fx * np.float32(2048))
    sx1 = np.minimum(sx + 1, w - 1)                                            # (a1 == 0 wherever this clamps)
    sy, fy = _coeffs(dh, h)
    b0 = _rnd_short((np.float32(1.0) - fy) * np.float32(2048)); b1 = _rnd_short(fy * np.float32(2048))
    sy0 = np.clip(sy, 0, h - 1); sy1 = np.clip(sy + 1, 0, h - 1)              # rows are clamped, weights are not
    src = img.astype(np.int64)
    hor = src[:, sx, :] * a0[None, :, None] + src[:, sx1, :] * a1[None, :, None]      # (h, dw, c) int32 range
    s0, s1 = hor[sy0], hor[sy1]
    out = (((b0[:, None, None] * (s0 >> 4)) >> 16) + ((b1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox(img, input_size, swap_rb):
    """-> ((3, H, W) float32, r): resize by r = min(H/h, W/w) to (int(w*r), int(h*r)), top-left aligned, pad 114"""
    h, w = img.shape[:2]
    H, W = input_size
    r = min(H / h, W / w)
    src = img[:, :, ::-1] if swap_rb else img
    rs = cv2_resize_linear_u8(np.ascontiguousarray(src), (int(w * r), int(h * r)))
    out = np.full((H, W, 3), 114, dtype=np.uint8)
    out[:int(h * r), :int(w * r)] = rs
    return np.ascontiguousarray(out.transpose(2, 0, 1), dtype=np.float32), r
