"""Epilogue-cost experiment: one GEMM shape with different epilogues / ablations.  usage: gemm_epi.py [cfg]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unicorn_amd import _lib as L
lib = L.lib()
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 0

def timeit(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

SHAPES = [(32000, 3072, 768), (32000, 768, 3072), (512000, 768, 192), (512000, 192, 768), (128000, 1536, 384), (128000, 1024, 256)]
if os.environ.get("EPI_SHAPES"): SHAPES = SHAPES[:int(os.environ["EPI_SHAPES"])]
for (M, N, K) in SHAPES:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    Npad, Kpad = (N + 255) // 256 * 256, (K + 63) // 64 * 64
    Wp = (torch.randn(Npad, Kpad, device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda")
    outF = torch.empty(M, N, device="cuda")
    outB = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    def run(act, use_res, use_F, use_B, dbg):
        return timeit(lambda: L.check(lib.uni_gemm_bf16(L.ptr(A), K, L.ptr(Wp), M, N, M, 1, K, 1, 1, 1, 0, L.ptr(bias), act,
                      L.ptr(res) if use_res else None, N, L.ptr(outF) if use_F else None, N, L.ptr(outB) if use_B else None, N,
                      None, 0, cfg + 1000 * dbg, L.stream_ptr()), "gemm"))
    fl = 2.0 * M * N * K
    row = []
    D = int(os.environ.get("EPI_DBG", "0"))   # 32 = direct (non-staged) stores
    for name, a in [("noepi", (0, 0, 0, 1, 16)), ("B", (0, 0, 0, 1, D)), ("B+gelu", (2, 0, 0, 1, D)), ("F", (0, 0, 1, 0, D)),
                    ("res+F+B", (0, 1, 1, 1, D))]:
        us = run(*a)
        row.append("%s %.0fus %.0fTF" % (name, us, fl / us / 1e6))
    print(M, N, K, " | ".join(row), flush=True)
