"""Micro-benchmark of the implicit-GEMM kernel over the shapes of the 800x1280 models (through the C-ABI)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from unicorn_amd import _lib as L

lib = L.lib()
SHAPES = [
    # (name, Hin, Win, Cin, N, k, stride)
    ("L.s0.pw1", 64000, 1, 192, 768, 1, 1), ("L.s0.pw2", 64000, 1, 768, 192, 1, 1),
    ("L.s1.pw1", 16000, 1, 384, 1536, 1, 1), ("L.s1.pw2", 16000, 1, 1536, 384, 1, 1),
    ("L.s2.pw1", 4000, 1, 768, 3072, 1, 1), ("L.s2.pw2", 4000, 1, 3072, 768, 1, 1),
    ("L.s3.pw1", 1000, 1, 1536, 6144, 1, 1), ("L.s3.pw2", 1000, 1, 6144, 1536, 1, 1),
    ("head.s8.3x3", 100, 160, 256, 256, 3, 1), ("head.s8.3x3x512", 100, 160, 256, 512, 3, 1),
    ("head.s16.3x3", 50, 80, 256, 256, 3, 1), ("head.s32.3x3", 25, 40, 256, 256, 3, 1),
    ("head.att.pw1", 16000, 1, 256, 1024, 1, 1), ("head.att.pw2", 16000, 1, 1024, 256, 1, 1),
    ("fpn.s8.1x1", 16000, 1, 768, 384, 1, 1), ("fpn.s8.3x3.192", 100, 160, 192, 192, 3, 1),
    ("up1", 100, 160, 64, 256, 3, 1), ("up3", 100, 160, 256, 128, 3, 1),
    ("T.s0.pw1", 64000, 1, 96, 384, 1, 1), ("T.s2.pw1", 4000, 1, 384, 1536, 1, 1),
]
import os as _os
SCALE = int(_os.environ.get("GEMM_SCALE", "1"))     # multiply M (emulates a batch of frames)
SHAPES = [(n, h * SCALE, w, c, nn, k, st) for (n, h, w, c, nn, k, st) in SHAPES]
cfgs = [int(a) for a in sys.argv[1:]] or [22, 21, 12, 11]
print("%-18s %8s %6s %6s | " % ("shape", "M", "N", "K") + " ".join("%9s" % ("cfg%d" % c) for c in cfgs))
for name, Hin, Win, Cin, N, k, stride in SHAPES:
    pad = (k - 1) // 2
    Hout, Wout = (Hin + 2 * pad - k) // stride + 1, (Win + 2 * pad - k) // stride + 1
    M, K = Hout * Wout, Cin * k * k
    A = (torch.randn(Hin * Win, Cin, device="cuda")).to(torch.bfloat16)
    Npad, Kpad = (N + 255) // 256 * 256, (K + 63) // 64 * 64
    Wp = (torch.randn(Npad, Kpad, device="cuda") * 0.05).to(torch.bfloat16)
    outB = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    res = []
    for cfg in cfgs:
        def run():
            L.check(lib.uni_gemm_bf16(L.ptr(A), Cin, L.ptr(Wp), M, N, Hin, Win, Cin, k, k, stride, pad, None, 0, None, 0, None, 0,
                                      L.ptr(outB), N, None, 0, cfg, L.stream_ptr()), "gemm")
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        res.append("%5.0fTF/%4.0fus" % (2.0 * M * N * K / ms / 1e9, ms * 1e3))
    print("%-18s %8d %6d %6d | " % (name, M, N, K) + " ".join(res), flush=True)
