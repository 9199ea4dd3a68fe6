"""Interaction stage (bottleneck + deformable encoder layer, MSDA inside) for B frame pairs at 800x1280; UNI_MSDA_V1=1 selects the
round-1 lane-per-channel sampler for A/B."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch, synth, unicorn_oracle as uo
from unicorn_amd.models import Unicorn
B = int(os.environ.get("B", "16"))
name = "unicorn_track_large"
m = Unicorn(name, precision=os.environ.get("PRECISION", "f16x2")).cuda(0); m.load_state_dict(synth.synth_state_dict(uo.CONFIGS[name]))
f = torch.randn(B, 768, 50, 80, device="cuda").contiguous(memory_format=torch.channels_last)
d0 = {"feat": f[:1].clone(), "pos": m._pos(50, 80), "h": 50, "w": 80}
d1 = {"feat": f, "pos": m._pos(50, 80).expand(B, -1, -1, -1), "h": 50, "w": 80}
for _ in range(3): m(seq_dict0=d0, seq_dict1=d1, mode="interaction")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): m(seq_dict0=d0, seq_dict1=d1, mode="interaction")
e1.record(); torch.cuda.synchronize()
print("interaction B=%d: %.3f ms per call (%s sampler)" % (B, e0.elapsed_time(e1) / 10, "lane-per-channel" if os.environ.get("UNI_MSDA_V1") else "wave-per-(token,head)"))
