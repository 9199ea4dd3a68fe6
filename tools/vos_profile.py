"""Where a VOS step spends its time: stage-by-stage wall clock (host-synchronised) for K objects, object-batched head vs loop."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch, synth, unicorn_oracle as uo
from unicorn_amd.models import Unicorn
from unicorn_amd.ops import corr_softmax_pv, label_map_s8, prior_pyramid
from unicorn_amd.utils.boxes import postprocess_inst
K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
name = "unicorn_track_large_mask"
cfg = uo.CONFIGS[name]
m = Unicorn(name, precision=os.environ.get("PRECISION", "f16x2")).cuda(0); m.load_state_dict(synth.synth_state_dict(cfg))
H, W = 800, 1280
frames, box = synth.synth_clip(H, W, 3, seed=1)
lbs = torch.cat([label_map_s8(torch.tensor([100.0 + 90 * i, 80.0 + 50 * i, 300.0 + 90 * i, 340.0 + 50 * i]), H, W, "cuda") for i in range(K)], 0)
def T(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
with torch.no_grad():
    _, d_pre = m(imgs=frames[0].cuda(), mode="backbone")
    cur = frames[1].cuda()
    t, (fpn, d_cur) = T(lambda: m(imgs=cur, mode="backbone")); print("backbone            %.2f ms" % t)
    t, (fp, fc) = T(lambda: m(seq_dict0=d_pre, seq_dict1=d_cur, mode="interaction")); print("interaction         %.2f ms" % t)
    t, ep = T(lambda: m(feat=fp, mode="upsample")); print("upsample (x1)       %.2f ms" % t)
    ec = m(feat=fc, mode="upsample")
    t, pred = T(lambda: corr_softmax_pv(ep.flatten(-2).squeeze(0), ec.flatten(-2).squeeze(0), lbs, precision=1)); print("corr K=%d            %.2f ms" % (K, t))
    pri = tuple(p.transpose(0, 1).contiguous() for p in prior_pyramid(pred.view(1, K, d_cur["h"] * 2, d_cur["w"] * 2)))
    t, ho = T(lambda: m.head(fpn, pri, mode="sot")); print("head batched K=%d    %.2f ms" % (K, t))
    t, _ = T(lambda: [m.head(fpn, tuple(p[k:k + 1] for p in pri), mode="sot") for k in range(K)]); print("head loop K=%d       %.2f ms" % (K, t))
    t, h1 = T(lambda: m.head(fpn, tuple(p[0:1] for p in pri), mode="sot")); print("head K=1            %.2f ms" % t)
    def post(h):
        o, l, d, lv, mf, um = h
        return postprocess_inst(o.clone(), l, d, lv, mf, m.head.mask_head, 1, 0.001, 0.65, d_rate=cfg.d_rate, up_masks=um, max_inst=1)
    t, _ = T(lambda: post(ho)); print("postprocess_inst K=%d %.2f ms" % (K, t))
    t, _ = T(lambda: post(h1)); print("postprocess_inst K=1 %.2f ms" % t)
