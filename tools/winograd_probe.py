"""GEMM halves of the Winograd F(2x2, 3x3) comparison (tools/winograd_probe.hip times the two transform passes): the engine's 3x3 implicit GEMM
(M x Cout x 9 Cin) against the 16 independent plain GEMMs (M / 4 x Cout x Cin, fp32 product planes out) Winograd would run, through the library's
own f16x2 kernels (launcher's tile choice).      python tools/winograd_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unicorn_amd import _lib as L

lib = L.lib()


def gemm(A, Wp, M, N, Hin, Win, Cin, k, outF):
    L.check(lib.uni_gemm_h2(L.ptr(A), Cin, L.ptr(Wp), 1.0 / 128, M, N, Hin, Win, Cin, k, k, 1, (k - 1) // 2, None, 0, None, N, L.ptr(outF), N, None, N,
                            None, 0, 0, L.stream_ptr()), "gemm_h2")


def timeit(fn, reps=5):
    best = 1e9
    for _ in range(3):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best * 1e3


for name, B, H, W, Cin, Cout in (("head tower 3x3 (s8)", 16, 100, 160, 256, 256), ("FPN 3x3 (s16)", 16, 50, 80, 384, 384), ("head tower 3x3, 1 frame", 1, 100, 160, 256, 256)):
    M = B * H * W

    def operand(rows, cols):
        x = torch.randn(rows, cols, device="cuda")
        a = torch.empty((rows, cols), device="cuda", dtype=torch.int32)
        L.check(lib.uni_cast_h2(L.ptr(x), cols, L.ptr(a), cols, rows, cols, L.stream_ptr()), "cast")
        return a
    # direct: implicit GEMM over the stacked maps
    A = operand(M, Cin)
    K9 = 9 * Cin
    W9 = operand((Cout + 255) // 256 * 256, (K9 + 63) // 64 * 64)
    out = torch.empty((M, Cout), device="cuda")
    t_direct = timeit(lambda: gemm(A, W9, M, Cout, B * H, W, Cin, 3, out))
    # Winograd: 16 plain GEMMs over the transformed planes
    V = [operand(M // 4, Cin) for _ in range(16)]
    W1 = [operand((Cout + 255) // 256 * 256, (Cin + 63) // 64 * 64) for _ in range(16)]
    Mo = [torch.empty((M // 4, Cout), device="cuda") for _ in range(16)]

    def wino():
        for p in range(16):
            gemm(V[p], W1[p], M // 4, Cout, M // 4, 1, Cin, 1, Mo[p])
    t_wino = timeit(wino)
    fl = 2.0 * M * Cout * K9
    print("%-24s direct 3x3 implicit GEMM %7.1f us (%5.1f TF-eq) | 16 plain GEMMs of M/4 x %d x %d: %7.1f us (%.2fx the direct time for 1/2.25 of its MFMAs)"
          % (name, t_direct, fl / t_direct / 1e6, Cout, Cin, t_wino, t_wino / t_direct))
