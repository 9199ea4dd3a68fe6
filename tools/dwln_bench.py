"""The fused depthwise 7x7 + LayerNorm kernels alone on the ConvNeXt / head maps of the large model at B frames, in the engine's operand
format (FMT: 2 = f16x2 default, 0 = bf16, 1 = fp32).  UNI_DW_NOLDSW=1 selects the non-persistent kernels, UNI_DW_PX / UNI_DW_ROWS the tiling."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unicorn_amd import _lib as L
lib = L.lib()
B = int(os.environ.get("B", "16"))
FMT = int(os.environ.get("FMT", "2"))
SHAPES = [(768, 50, 80), (192, 200, 320), (384, 100, 160), (256, 100, 160), (256, 50, 80), (1536, 25, 40), (256, 25, 40)]
if os.environ.get("SHAPES"):      # e.g. SHAPES=1536x25x40,256x25x40
    SHAPES = [tuple(int(v) for v in t.split("x")) for t in os.environ["SHAPES"].split(",")]
for (C, H, W) in SHAPES:
    x = torch.randn(B, H, W, C, device="cuda")
    w = torch.randn(49, C, device="cuda") * 0.1
    b, g, be = torch.randn(C, device="cuda"), torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
    out = torch.empty((B * H * W, C), device="cuda", dtype=torch.bfloat16 if FMT == 0 else torch.float32)
    def run():
        L.check(lib.uni_dwconv7_ln_ex(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(g), L.ptr(be), 1e-6, B, H, W, C, L.ptr(out), FMT, L.stream_ptr()), "dwln_ex")
    best = 1e9
    for rep in range(3):
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    eb = 6 if FMT == 0 else 8
    print("C=%4d %3dx%3d B=%d: %7.1f us  %.2f TB/s algorithmic (%d B/elem)" % (C, H, W, B, best * 1e3, B * H * W * C * eb / best / 1e9, eb))
