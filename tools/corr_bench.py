import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unicorn_amd.ops import corr_softmax_pv
a = torch.randn(128, 16000, device="cuda") * 0.5; b = torch.randn(128, 16000, device="cuda") * 0.5; v = torch.rand(1, 16000, device="cuda")
for _ in range(3): corr_softmax_pv(a, b, v)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): corr_softmax_pv(a, b, v)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(os.environ.get("UNI_CORR_BLOCKS"), "ms", round(ms, 4), "TF", round(2 * 16000 * 16000 * 128 / ms / 1e9, 1))
