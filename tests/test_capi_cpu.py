"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/unicorn_hip.h declares; host-side packing helpers agree with numpy; loud failure without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "unicorn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(uni_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from unicorn_amd import _lib
    lib = _lib.lib()
    syms = _declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), "libunicorn_hip.so does not export %s" % s
    assert set(syms) == set(_lib.PROTOS), set(syms) ^ set(_lib.PROTOS)
    assert lib.uni_version() == 1


def test_pack_weight_matches_numpy():
    from unicorn_amd import _lib
    lib = _lib.lib()
    g = torch.Generator().manual_seed(0)
    w = torch.randn(5, 16, 3, 3, generator=g)
    out = np.zeros((256, 192), dtype=np.uint16)
    wc = np.ascontiguousarray(w.numpy())
    assert lib.uni_pack_weight(wc.ctypes.data_as(C.c_void_p), 5, 16, 3, 3, out.ctypes.data_as(C.c_void_p)) == 0
    exp = w.permute(0, 2, 3, 1).reshape(5, 144).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(out[:5, :144], exp)
    assert not out[5:].any() and not out[:, 144:].any()


def test_pack_weight_h2_matches_numpy():
    """The host packer of the f16x2 operand format (the headline precision): per 8 consecutive k a 32-byte group
    [8 x hi][8 x lo] with hi = f16(x * scale), lo = f16(x * scale - hi), one power-of-two scale per tensor that puts max |w|
    into [2^14, 2^15); k = (ky * KW + kx) * Cin + c; rows / columns beyond N / K are zero."""
    from unicorn_amd import _lib
    lib = _lib.lib()
    g = torch.Generator().manual_seed(1)
    N, Cin, k = 5, 16, 3
    w = torch.randn(N, Cin, k, k, generator=g) * 0.03
    w[0, 0, 0, 0] = 0.0
    w[1, 2, 1, 1] = 1e-7                                   # its lo half is an f16 subnormal / zero: must not blow up
    K = Cin * k * k
    Npad, Kpad = 256, (K + 63) // 64 * 64
    out = np.zeros((Npad, Kpad), dtype=np.uint32)
    wc = np.ascontiguousarray(w.numpy())
    sc = C.c_float(0)
    assert lib.uni_pack_weight_h2(wc.ctypes.data_as(C.c_void_p), N, Cin, k, k, out.ctypes.data_as(C.c_void_p), C.byref(sc)) == 0
    inv = sc.value                                         # the accumulator multiplier = 1 / scale
    scale = 1.0 / inv
    assert np.log2(scale) == round(np.log2(scale)) and 2.0 ** 14 <= float(w.abs().max()) * scale < 2.0 ** 15
    x = (w.permute(0, 2, 3, 1).reshape(N, K).numpy().astype(np.float32) * np.float32(scale))
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    halves = out.view(np.uint16).reshape(Npad, Kpad // 8, 16)      # [row][group of 8 k][8 hi | 8 lo]
    got_hi = halves[:N, :, :8].reshape(N, -1)[:, :K].view(np.float16)
    got_lo = halves[:N, :, 8:].reshape(N, -1)[:, :K].view(np.float16)
    assert np.array_equal(got_hi.view(np.uint16), hi.view(np.uint16))
    assert np.array_equal(got_lo.view(np.uint16), lo.view(np.uint16))
    rec = (got_hi.astype(np.float64) + got_lo.astype(np.float64)) * inv
    ref = w.permute(0, 2, 3, 1).reshape(N, K).numpy().astype(np.float64)
    assert np.abs(rec - ref).max() <= np.abs(ref).max() * 2.0 ** -21
    assert not out[N:].any() and not halves[:N].reshape(N, -1, 16)[:, (K + 7) // 8:].any()


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_fails_loudly_without_gpu():
    from unicorn_amd import _lib
    from unicorn_amd.models import Unicorn
    from unicorn_amd.ops import corr_softmax_pv
    m = Unicorn("unicorn_track_tiny")
    with pytest.raises(_lib.UnicornHipError):
        m.cuda()
    with pytest.raises(_lib.UnicornHipError):
        corr_softmax_pv(torch.zeros(128, 4), torch.zeros(128, 4), torch.zeros(1, 4))
    with pytest.raises(_lib.UnicornHipError):
        m(imgs=torch.zeros(1, 3, 32, 32), mode="backbone")
