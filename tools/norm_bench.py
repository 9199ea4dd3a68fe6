"""Micro-benchmark of the HBM-bound kernels at the shapes of the 800x1280 large model."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unicorn_amd import _lib as L
lib = L.lib()

def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

SC = int(os.environ.get("NORM_SCALE", "1"))
print("dwconv7_ln  (H, W, C): us, GB/s (algorithmic 6 B/elem)")
for H, W, C in [(h * SC, w, c) for (h, w, c) in [(200, 320, 192), (100, 160, 384), (50, 80, 768), (25, 40, 1536), (100, 160, 256), (50, 80, 256), (25, 40, 256), (200, 320, 96)]]:
    x = torch.randn(H * W, C, device="cuda"); w = torch.randn(49, C, device="cuda"); b = torch.randn(C, device="cuda")
    g = torch.randn(C, device="cuda"); be = torch.randn(C, device="cuda"); out = torch.empty(H * W, C, device="cuda", dtype=torch.bfloat16)
    us = timeit(lambda: L.check(lib.uni_dwconv7_ln(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(g), L.ptr(be), 1e-6, H, W, C, L.ptr(out), L.stream_ptr()), "dw"))
    print("  %4d %4d %5d : %7.1f us  %7.0f GB/s" % (H, W, C, us, H * W * C * 6 / us / 1e3))
print("gn_apply (M, C): us, GB/s (fp32 in, bf16 out = 6 B/elem)")
for M, C in [(16000, 256), (16000, 512), (16000, 384), (4000, 768), (4000, 256), (1000, 256), (1000, 1536), (64000, 192)]:
    x = torch.randn(M, C, device="cuda"); st = torch.rand(64, device="cuda", dtype=torch.float64) * M
    g = torch.randn(C, device="cuda"); be = torch.randn(C, device="cuda"); out = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
    us = timeit(lambda: L.check(lib.uni_groupnorm_act(L.ptr(x), L.ptr(st), L.ptr(g), L.ptr(be), 1e-3, M, C, 16, 3, None, L.ptr(out), L.stream_ptr()), "gn"))
    print("  %6d %5d : %7.1f us  %7.0f GB/s" % (M, C, us, M * C * 6 / us / 1e3))
print("layernorm (M, C): fp32 in, bf16 out")
for M, C in [(64000, 192), (16000, 384), (4000, 768), (8000, 256)]:
    x = torch.randn(M, C, device="cuda"); g = torch.randn(C, device="cuda"); be = torch.randn(C, device="cuda")
    out = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
    us = timeit(lambda: L.check(lib.uni_layernorm(L.ptr(x), C, L.ptr(g), L.ptr(be), 1e-6, M, C, None, L.ptr(out), L.stream_ptr()), "ln"))
    print("  %6d %5d : %7.1f us  %7.0f GB/s" % (M, C, us, M * C * 6 / us / 1e3))
