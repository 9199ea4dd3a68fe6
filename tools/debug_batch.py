import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import torch, synth, unicorn_oracle as uo
from unicorn_amd.models import Unicorn
from unicorn_amd.ops import corr_softmax_pv, label_map_s8, prior_pyramid
name = sys.argv[1]; B = int(sys.argv[2]); H, W = 800, 1280
cfg = uo.CONFIGS[name]; P = synth.synth_state_dict(cfg)
m = Unicorn(name).cuda(); m.load_state_dict(P)
x = torch.rand(B, 3, H, W, device="cuda") * 255
def sync(tag):
    torch.cuda.synchronize(); print("ok", tag, flush=True)
with torch.no_grad():
    _, dp = m(imgs=x[:1], mode="backbone"); sync("ref backbone")
    fpn, d = m(imgs=x, mode="backbone"); sync("backbone")
    fp, fc = m(seq_dict0=dp, seq_dict1=d, mode="interaction"); sync("interaction")
    ep = m(feat=fp, mode="upsample"); ec = m(feat=fc, mode="upsample"); sync("upsample")
    lbs = label_map_s8([320., 200., 640., 400.], H, W, "cuda")
    pred = torch.cat([corr_softmax_pv(ep[b].flatten(-2), ec[b].flatten(-2), lbs) for b in range(B)], 0); sync("corr")
    pri = tuple(t.transpose(0, 1).contiguous() for t in prior_pyramid(pred.view(1, B, H // 8, W // 8))); sync("pyramid")
    out = m.head(fpn, pri, mode="sot"); sync("head")
    print(out.shape, float(out.abs().mean()))
