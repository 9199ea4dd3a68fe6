import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import torch, torch.nn.functional as F
from unicorn_amd import _lib as L
def run(C_, B, H, W, fmt, variant):
    g = torch.Generator().manual_seed(C_ + H + B)
    x = torch.randn(B, C_, H, W, generator=g)
    w = torch.randn(C_, 1, 7, 7, generator=g) / 7
    b, ga, be = torch.randn(C_, generator=g) * 0.1, 1 + 0.1 * torch.randn(C_, generator=g), 0.1 * torch.randn(C_, generator=g)
    y = F.conv2d(x, w, b, padding=3, groups=C_).permute(0, 2, 3, 1)
    exp = F.layer_norm(y, (C_,), ga, be, 1e-6).reshape(-1, C_)
    xn = x.permute(0, 2, 3, 1).contiguous().cuda(); wt = w.reshape(C_, 49).t().contiguous().cuda()
    M = B * H * W
    out = torch.full((M, C_), 7.0, device="cuda", dtype=torch.bfloat16 if fmt == 0 else torch.float32)
    bd, gd, bed = b.cuda(), ga.cuda(), be.cuda()
    rc = L.lib().uni_dwconv7_ln_ex(L.ptr(xn), L.ptr(wt), L.ptr(bd), L.ptr(gd), L.ptr(bed), 1e-6, B, H, W, C_, L.ptr(out), fmt, variant, L.stream_ptr())
    torch.cuda.synchronize()
    if fmt == 2:
        gq = out.view(torch.float16).reshape(M, C_ // 8, 16).float().cpu(); o = (gq[:, :, :8] + gq[:, :, 8:]).reshape(M, C_)
    else:
        o = out.float().cpu()
    d = (o - exp).abs()
    bad = torch.nonzero(~(d < 1e-3))
    print("C=%d B=%d %dx%d fmt=%d variant=%d rc=%d: bad=%d of %d, max(ok)=%.2e" % (C_, B, H, W, fmt, variant, rc, bad.shape[0], d.numel(), float(d[d < 1e-3].max()) if bool((d < 1e-3).any()) else -1.0))
    if bad.shape[0]:
        rows = bad[:, 0]; b_ = rows // (H * W); yy = (rows % (H * W)) // W; xx = rows % W
        print("  samples", sorted(set(b_.tolist()))[:8], "rows y", sorted(set(yy.tolist()))[:20], "x", sorted(set(xx.tolist()))[:20], "ch", sorted(set(bad[:, 1].tolist()))[:12], "n-ch", len(set(bad[:,1].tolist())))
        print("  sample values", o[bad[0,0], bad[0,1]].item(), exp[bad[0,0], bad[0,1]].item())
for fmt in (1, 2):
    for v in (1, 2):
        run(768, 6, 49, 83, fmt, v)
run(256, 3, 57, 90, 1, 1)
run(192, 2, 101, 163, 1, 1)
