"""Run ONE implicit-GEMM shape N times (for rocprofv3 --pmc passes).  usage: gemm_one.py Hin Win Cin N k stride cfg reps"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unicorn_amd import _lib as L
lib = L.lib()
Hin, Win, Cin, N, k, stride, cfg, reps = [int(a) for a in sys.argv[1:9]]
pad = (k - 1) // 2
Hout, Wout = (Hin + 2 * pad - k) // stride + 1, (Win + 2 * pad - k) // stride + 1
M, K = Hout * Wout, Cin * k * k
A = torch.randn(Hin * Win, Cin, device="cuda").to(torch.bfloat16)
Npad, Kpad = (N + 255) // 256 * 256, (K + 63) // 64 * 64
Wp = (torch.randn(Npad, Kpad, device="cuda") * 0.05).to(torch.bfloat16)
outB = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
for _ in range(reps):
    L.check(lib.uni_gemm_bf16(L.ptr(A), Cin, L.ptr(Wp), M, N, Hin, Win, Cin, k, k, stride, pad, None, 0, None, 0, None, 0,
                              L.ptr(outB), N, None, 0, cfg, L.stream_ptr()), "gemm")
torch.cuda.synchronize()
print("done", M, N, K)
