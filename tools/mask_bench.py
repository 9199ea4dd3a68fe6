"""CondInst masks of N candidates at the bench geometry (mask_feats 100 x 160, up_rate 4, d_rate 2 -> 800 x 1280; 1080p image, r = 2/3):
two passes through the (N, 800, 1280) fp32 maps (uni_condinst_masks + uni_mask_resize) against the fused entry point
(uni_condinst_masks_u8), and the RLE encoder on the resulting masks.   python tools/mask_bench.py [N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unicorn_amd.ops import condinst_masks, condinst_masks_resized, mask_resize, mots_overlap_free, rle_encode

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator().manual_seed(0)
H8, W8, H, W, r = 100, 160, 1080, 1920, 800 / 1200
mf = torch.randn(1, 8, H8, W8, generator=g).cuda()
um = torch.randn(1, 144, H8, W8, generator=g).cuda()
params = (torch.randn(N, 169, generator=g) * 0.5).cuda()
loc = (torch.rand(N, 2, generator=g) * torch.tensor([W8 * 8.0, H8 * 8.0])).cuda()
lvl = torch.randint(0, 5, (N,), generator=g)


def T(fn, n=10):
    for _ in range(3):
        out = fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, out


t2, ref = T(lambda: mask_resize(condinst_masks(mf, um, params, loc, lvl, 4, 2)[:, 0], r, H, W, thr=0.3))
t1, got = T(lambda: condinst_masks_resized(mf, um, params, loc, lvl, 4, 2, r, H, W, thr=0.3))
print("%d candidates: two passes %.1f us, fused %.1f us, identical %s (mask density %.3f)" % (N, t2, t1, bool(torch.equal(ref, got)), float(got.float().mean())))
free = mots_overlap_free(got)
t3, _ = T(lambda: rle_encode(free))
print("overlap-free + rle_encode of %d masks (host-synchronised, strings to the host): %.1f us" % (N, t3))
