"""The LDS-tiled raw depthwise kernel (dwconv.hip) alone on the ConvNeXt shapes of the large model at 16 frames.
UNI_DW_DBG ablations: 1 no DMA after chunk 0, 2 one tap row only, 4 no output stores."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unicorn_amd import _lib as L
lib = L.lib()
B = int(os.environ.get("B", "16"))
for (C, H, W) in [(768, 50, 80), (192, 200, 320), (384, 100, 160), (1536, 25, 40), (256, 100, 160)]:
    x = torch.randn(B, H, W, C, device="cuda")
    w = torch.randn(49, C, device="cuda") * 0.1
    b = torch.randn(C, device="cuda")
    out = torch.empty((B * H * W, C), device="cuda", dtype=torch.int32)
    st = torch.empty((B * H * W, 2), device="cuda")
    def run():
        L.check(lib.uni_dwconv7_raw(L.ptr(x), L.ptr(w), L.ptr(b), 1e-6, B, H, W, C, 2, L.ptr(out), L.ptr(st), L.stream_ptr()), "dw")
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("C=%4d %3dx%3d B=%d: %7.1f us  %.2f TB/s algorithmic (8 B/elem)" % (C, H, W, B, ms * 1e3, B * H * W * C * 8 / ms / 1e9))
