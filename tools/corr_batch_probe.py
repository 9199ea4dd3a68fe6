"""Upper bound of batching the 16 per-frame correlations of a step into one launch: 16 calls with Q = 16000 against ONE call with Q = 256000
(same reference side for all frames: slightly optimistic, the real batch has a reference embedding per frame)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unicorn_amd.ops import corr_softmax_pv
B = 16
a = (torch.randn(16000, 128, device="cuda") * 0.5).t(); v = torch.rand(1, 16000, device="cuda")
bs = (torch.randn(B * 16000, 128, device="cuda") * 0.5)
per = [bs[i * 16000:(i + 1) * 16000].t() for i in range(B)]
allq = bs.t()
def t(fn, n=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for rep in range(2):
    print("16 calls  : %.3f ms per frame" % (t(lambda: [corr_softmax_pv(a, p, v, precision=2) for p in per]) / B))
    print("one call  : %.3f ms per frame" % (t(lambda: corr_softmax_pv(a, allq, v, precision=2)) / B))
