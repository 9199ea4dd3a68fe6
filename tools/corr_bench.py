"""Correlation kernel alone at 800x1280 scale (R = Q = 16000, C = 128).  env UNI_CORR_BLOCKS overrides the block target."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unicorn_amd.ops import corr_softmax_pv
a = torch.randn(128, 16000, device="cuda") * 0.5; b = torch.randn(128, 16000, device="cuda") * 0.5; v = torch.rand(1, 16000, device="cuda")
for prec in (0, 1, 2):
    for _ in range(3): corr_softmax_pv(a, b, v, precision=prec)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): corr_softmax_pv(a, b, v, precision=prec)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("precision", prec, "blocks", os.environ.get("UNI_CORR_BLOCKS"), "ms", round(ms, 4), "TF(fp32-equivalent)", round(2 * 16000 * 16000 * 128 / ms / 1e9, 1))
