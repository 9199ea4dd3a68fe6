"""Single-frame (B = 1) GEMM shapes of the 800x1280 large model: tile configuration x split-K sweep of uni_gemm_h2.
    python tools/gemm_b1_bench.py      prints us per (cfg, splitk); cfg 0 = launcher heuristic"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unicorn_amd import _lib as L

lib = L.lib()
SHAPES = [
    # (name, Hin, Win, Cin, N, k, act, res(in place), outB)
    ("s2.pw2 4000x768x3072", 4000, 1, 3072, 768, 1, 0, 1, 0), ("s2.pw1 4000x3072x768 gelu", 4000, 1, 768, 3072, 1, 2, 0, 1),
    ("fpn3x3 4000x384x3456", 50, 80, 384, 384, 3, 0, 0, 0), ("fpn3x3 1000x768x6912", 25, 40, 768, 768, 3, 0, 0, 0),
    ("head3x3 16000x256x2304", 100, 160, 256, 256, 3, 0, 0, 0), ("head3x3 4000x256x2304", 50, 80, 256, 256, 3, 0, 0, 0),
    ("head3x3 1000x256x2304", 25, 40, 256, 256, 3, 0, 0, 0),
    ("s3.pw2 1000x1536x6144", 1000, 1, 6144, 1536, 1, 0, 1, 0), ("s3.pw1 1000x6144x1536 gelu", 1000, 1, 1536, 6144, 1, 2, 0, 1),
    ("s1.pw2 16000x384x1536", 16000, 1, 1536, 384, 1, 0, 1, 0), ("s1.pw1 16000x1536x384 gelu", 16000, 1, 384, 1536, 1, 2, 0, 1),
    ("fpn1x1 4000x768x1536", 4000, 1, 1536, 768, 1, 0, 0, 0), ("fpn1x1 1000x1536x3072", 1000, 1, 3072, 1536, 1, 0, 0, 0),
    ("fpn3x3 16000x192x1728", 100, 160, 192, 192, 3, 0, 0, 0), ("head3x3 16000x128x2304", 100, 160, 256, 128, 3, 0, 0, 0),
    ("1x1 4000x384x384", 4000, 1, 384, 384, 1, 0, 0, 0), ("1x1 16000x256x1024", 16000, 1, 1024, 256, 1, 0, 0, 0),
    ("1x1 16000x1024x256", 16000, 1, 256, 1024, 1, 0, 0, 0), ("1x1 4000x768x768", 4000, 1, 768, 768, 1, 0, 0, 0),
    ("1x1 16000x192x192", 16000, 1, 192, 192, 1, 0, 0, 0), ("1x1 4000x256x1024", 4000, 1, 1024, 256, 1, 0, 0, 0),
]
CFGS = [int(v) for v in os.environ.get("CFGS", "0,188,22,11,322,323,332,331,422,423").split(",")]
SPLITS = [int(v) for v in os.environ.get("SPLITS", "1,2,3,4").split(",")]
SPLIT_CFGS = tuple(int(v) for v in os.environ.get("SPLIT_CFGS", "188,22,323,331").split(","))
if os.environ.get("ONLY"):
    SHAPES = [s_ for s_ in SHAPES if any(t in s_[0] for t in os.environ["ONLY"].split(","))]
print("%-28s | " % "shape" + " ".join("%9s" % ("c%d/s%d" % (c, sk)) for c in CFGS for sk in (SPLITS if c in SPLIT_CFGS else [1])))
for name, Hin, Win, Cin, N, k, act, use_res, use_B in SHAPES:
    pad = (k - 1) // 2
    M = Hin * Win
    K = Cin * k * k
    x = torch.randn(M, Cin, device="cuda")
    A = torch.empty((M, Cin), device="cuda", dtype=torch.int32)
    L.check(lib.uni_cast_h2(L.ptr(x), Cin, L.ptr(A), Cin, M, Cin, L.stream_ptr()), "cast")
    Npad, Kpad = (N + 255) // 256 * 256, (K + 63) // 64 * 64
    wf = torch.randn(Npad, Kpad, device="cuda") * 100.0
    Wp = torch.empty((Npad, Kpad), device="cuda", dtype=torch.int32)
    L.check(lib.uni_cast_h2(L.ptr(wf), Kpad, L.ptr(Wp), Kpad, Npad, Kpad, L.stream_ptr()), "cast")
    bias = torch.randn(N, device="cuda")
    outF = torch.zeros((M, N), device="cuda")
    outB = torch.empty((M, N), device="cuda", dtype=torch.int32) if use_B else None
    cells = []
    for cfg in CFGS:
        for sk in (SPLITS if cfg in SPLIT_CFGS else [1]):
            if sk > 1 and (act or use_B):
                cells.append("        -")
                continue
            res = outF if use_res else None

            def run():
                L.check(lib.uni_gemm_h2(L.ptr(A), Cin, L.ptr(Wp), 1.0 / 128, M, N, Hin, Win, Cin, k, k, 1, pad, L.ptr(bias), act, L.ptr(res), N,
                                        None if use_B else L.ptr(outF), N, L.ptr(outB), N, None, 0, cfg + (1000000 * sk if sk > 1 else 0), L.stream_ptr()), "gemm_h2")
            try:
                for _ in range(3):
                    run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run()
                e1.record()
                torch.cuda.synchronize()
                cells.append("%9.1f" % (e0.elapsed_time(e1) / 10 * 1e3))
            except Exception as ex:      # a configuration the launcher rejects for this problem
                cells.append("      n/a")
    print("%-28s | " % name + " ".join(cells), flush=True)
