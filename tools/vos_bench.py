"""VOS step (BASELINE config 3: unicorn_track_large_mask + CondInst masks) with K objects, object-batched (row N3) vs the
reference's per-object head loop, same kernels.  usage: vos_bench.py [K]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch, synth, unicorn_oracle as uo
from unicorn_amd.models import Unicorn
from unicorn_amd.tracker import UnicornVOSTrack
K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
name = "unicorn_track_large_mask"
cfg = uo.CONFIGS[name]
m = Unicorn(name, precision=os.environ.get("PRECISION", "f16x2")).cuda(0); m.load_state_dict(synth.synth_state_dict(cfg))
H, W = 800, 1280
frames, box = synth.synth_clip(H, W, 3, seed=1)
ids = [str(i + 1) for i in range(K)]
boxes = {k: [100.0 + 90 * i, 80.0 + 50 * i, 200.0, 260.0] for i, k in enumerate(ids)}
trk = UnicornVOSTrack(m, input_size=(H, W), d_rate=cfg.d_rate, object_batched=True)
trk.initialize(frames[0].cuda(), {"init_object_ids": ids, "init_bbox": boxes})
trl = UnicornVOSTrack(m, input_size=(H, W), d_rate=cfg.d_rate, object_batched=False)     # same hoisted correlation, head once per object
trl.initialize(frames[0].cuda(), {"init_object_ids": ids, "init_bbox": boxes})
cur = frames[1].cuda()
def batched(): trk.step(cur)
def looped(): trl.step(cur)
MODE = os.environ.get("MODE", "")
for name_, fn in (("object-batched", batched), ("per-object loop", looped)):
    if MODE and MODE not in name_:
        continue
    its = []
    for _ in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); its.append((time.perf_counter() - t0) * 1e3)
    print("   per-iteration ms:", " ".join("%.1f" % v for v in its))
    dt = sorted(its[3:])[len(its[3:]) // 2] / 1e3
    print("VOS step K=%d %-16s %.2f ms/frame (%.1f fps)" % (K, name_, dt * 1e3, 1 / dt))
