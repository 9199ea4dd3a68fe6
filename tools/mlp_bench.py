"""Fused ConvNeXt MLP (mlp_fused.hip) vs the two launches it replaces (uni_gemm_h2: pwconv1 + GELU, pwconv2 + residual) on the
shapes of the 800x1280 large model at GEMM_SCALE frames per launch.  TF = fp32-equivalent TFLOP/s over both GEMMs.
    python tools/mlp_bench.py [dbg ...]        dbg: 0 full kernel, 1 no DMA, 2 no MFMA (C = 192 only)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from unicorn_amd import _lib as L

lib = L.lib()
SCALE = int(os.environ.get("GEMM_SCALE", "16"))
SHAPES = [("stage0 C=192", 192, 64000), ("head.l0 C=256", 256, 16000), ("head.l1 C=256", 256, 4000), ("tiny.s0 C=96", 96, 64000),
          ("tiny.s1 C=192", 192, 16000)]
dbgs = [int(a) for a in sys.argv[1:]] or [0]


def timeit(fn, n=6, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, Cc, m1 in SHAPES:
    M = m1 * SCALE
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, Cc, device="cuda")
    w1, w2 = torch.randn(4 * Cc, Cc, generator=g) * 0.05, torch.randn(Cc, 4 * Cc, generator=g) * 0.05
    b1, b2, gamma = torch.randn(4 * Cc, generator=g) * 0.1, torch.randn(Cc, generator=g) * 0.1, torch.rand(Cc, generator=g) + 0.5
    A = torch.empty((M, Cc), device="cuda", dtype=torch.int32)
    L.check(lib.uni_cast_h2(L.ptr(x), Cc, L.ptr(A), Cc, M, Cc, L.stream_ptr()), "cast")
    blobs = {}
    s1, s2 = C.c_float(0), C.c_float(0)
    for lay in ((0, 1) if Cc in (192, 256) else (0,)):
        blob = np.zeros(lib.uni_mlp_blob_bytes(Cc) // 2, dtype=np.uint16)
        L.check(lib.uni_mlp_pack(w1.numpy().ctypes.data_as(C.c_void_p), w2.numpy().ctypes.data_as(C.c_void_p), gamma.numpy().ctypes.data_as(C.c_void_p),
                                 Cc, lay, blob.ctypes.data_as(C.c_void_p), C.byref(s1), C.byref(s2)), "pack")
        blobs[lay] = torch.from_numpy(blob.view(np.int16)).cuda()
    b1d, b2d = b1.cuda(), (gamma * b2).cuda()
    res = torch.randn(M, Cc, device="cuda")
    out = torch.empty_like(res)
    # unfused pair
    def packh2(w):
        N, K = w.shape
        o = np.zeros(((N + 255) // 256 * 256, (K + 63) // 64 * 64), dtype=np.uint32)
        sc = C.c_float(0)
        L.check(lib.uni_pack_weight_h2(np.ascontiguousarray(w.numpy()).ctypes.data_as(C.c_void_p), N, K, 1, 1, o.ctypes.data_as(C.c_void_p), C.byref(sc)), "pack_h2")
        return torch.from_numpy(o.view(np.int32)).cuda(), sc.value
    W1p, q1 = packh2(w1)
    W2p, q2 = packh2(gamma[:, None] * w2)
    hid = torch.empty((M, 4 * Cc), device="cuda", dtype=torch.int32)
    out2 = torch.empty_like(res)

    def unfused():
        L.check(lib.uni_gemm_h2(L.ptr(A), Cc, L.ptr(W1p), q1, M, 4 * Cc, M, 1, Cc, 1, 1, 1, 0, L.ptr(b1d), 2, None, 0, None, 0, L.ptr(hid), 4 * Cc,
                                None, 0, 0, L.stream_ptr()), "pw1")
        L.check(lib.uni_gemm_h2(L.ptr(hid), 4 * Cc, L.ptr(W2p), q2, M, Cc, M, 1, 4 * Cc, 1, 1, 1, 0, L.ptr(b2d), 0, L.ptr(res), Cc, L.ptr(out2), Cc,
                                None, 0, None, 0, 0, L.stream_ptr()), "pw2")

    flop = 2.0 * 2.0 * M * 4 * Cc * Cc
    t_un = timeit(unfused)
    line = "%-16s M=%8d | unfused %7.1f us %6.1f TF |" % (name, M, t_un * 1e3, flop / t_un / 1e9)
    for lay in sorted(blobs):
        for d in dbgs:
            if d and Cc != 192:
                continue
            def fused():
                L.check(lib.uni_mlp_fused(L.ptr(A), Cc, L.ptr(blobs[lay]), L.ptr(b1d), L.ptr(b2d), s1.value, s2.value, L.ptr(res), Cc, L.ptr(out), Cc,
                                          None, 0, M, Cc, lay, d, L.stream_ptr()), "mlp_fused")
            t = timeit(fused)
            line += " L%d[dbg%d] %7.1f us %6.1f TF |" % (lay, d, t * 1e3, flop / t / 1e9)
            if d == 0:
                torch.cuda.synchronize()
                err = (out - out2).abs().max().item()
                line += " maxdiff %.2e |" % err
    print(line, flush=True)
