"""PCIe-inclusive note for DESIGN.md: time to hand one step's frames (B,3,800,1280) fp32 from pinned host memory to HBM."""
import torch, time
B = 16
h = torch.empty((B, 3, 800, 1280), dtype=torch.float32).pin_memory()
d = torch.empty_like(h, device="cuda")
u8 = torch.empty((B, 800, 1280, 3), dtype=torch.uint8).pin_memory()
d8 = torch.empty_like(u8, device="cuda")
for name, src, dst in (("fp32 CHW", h, d), ("uint8 HWC", u8, d8)):
    for _ in range(2): dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("%s: %.3f ms per %d frames = %.3f ms/frame, %.1f GB/s" % (name, dt * 1e3, B, dt * 1e3 / B, src.numel() * src.element_size() / dt / 1e9))
