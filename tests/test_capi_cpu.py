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
