"""K independent video streams on ONE MI355X, each calling the per-frame SOT step ONE FRAME PER CALL and synchronising on its own HIP stream
per frame -- the reference harness's own pattern: `tools/test.py ... --threads 32` runs many sequences per GPU in worker processes
(external/lib/test/evaluation/running.py:111-120), each a one-frame-per-call driver (unicorn_sot.py:57-108).  One frame per call leaves the chip
half empty (every layer is ONE round of tiles with its fill and drain, GEMM class at 0.27 of the format peak); concurrent streams fill those
holes with another stream's kernels.  Here: K python threads in one process, one context (weights + workspace) and one HIP stream each.
    python tools/concurrent_streams.py [--streams 1,2,4,8] [--frames 30] [--model unicorn_track_large]
Prints per K: aggregate frames/s, mean per-call latency."""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))      # synthetic weights / clip generator only (no oracle arithmetic is executed here)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", default="1,2,4,8")
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--model", default="unicorn_track_large")
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch
    import synth
    import unicorn_oracle as uo
    from unicorn_amd.models import Unicorn
    from unicorn_amd.ops import corr_softmax_pv, label_map_s8, prior_pyramid
    dev = torch.device("cuda:0")
    H, W = args.height, args.width
    cfg = uo.CONFIGS[args.model]
    P = synth.synth_state_dict(cfg)
    ks = [int(v) for v in args.streams.split(",")]
    kmax = max(ks)
    frames, box = synth.synth_clip(H, W, 5, seed=1)
    frames = [f.to(dev) for f in frames]
    lbs = label_map_s8(box, H, W, dev)

    class Tracker:
        def __init__(self):
            self.m = Unicorn(args.model, precision="f16x2").cuda(0)
            self.m.load_state_dict(P)
            self.s = torch.cuda.Stream(device=dev)
            with torch.no_grad(), torch.cuda.stream(self.s):
                _, self.d_pre = self.m(imgs=frames[0], mode="backbone")
            self.s.synchronize()
            self.lat = []

        def step(self, img):
            m = self.m
            fpn, d_cur = m(imgs=img, mode="backbone")
            f_pre, f_cur = m(seq_dict0=self.d_pre, seq_dict1=d_cur, mode="interaction")
            e_pre, e_cur = m(feat=f_pre, mode="upsample"), m(feat=f_cur, mode="upsample")
            pred = corr_softmax_pv(e_pre[0].flatten(-2), e_cur[0].flatten(-2), lbs)
            coarse = pred.view(1, 1, d_cur["h"] * 2, d_cur["w"] * 2)
            return m.head(fpn, prior_pyramid(coarse), mode="sot")

        def run(self, n, barrier):
            with torch.no_grad(), torch.cuda.stream(self.s):
                for i in range(3):
                    self.step(frames[1 + i % 4])
                self.s.synchronize()
                barrier.wait()
                self.lat = []
                for i in range(n):
                    t0 = time.perf_counter()
                    out = self.step(frames[1 + i % 4])
                    self.s.synchronize()                      # the driver reads the box back every frame
                    self.lat.append(time.perf_counter() - t0)
                self.out = out

    trk = [Tracker() for _ in range(kmax)]
    res = []
    for k in ks:
        bar = threading.Barrier(k + 1)
        th = [threading.Thread(target=trk[j].run, args=(args.frames, bar)) for j in range(k)]
        for t in th:
            t.start()
        bar.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        lat = [v for j in range(k) for v in trk[j].lat]
        row = {"streams": k, "frames_per_stream": args.frames, "aggregate_fps": round(k * args.frames / dt, 2),
               "latency_ms_mean": round(1e3 * sum(lat) / len(lat), 3), "latency_ms_max": round(1e3 * max(lat), 3)}
        res.append(row)
        print(json.dumps(row), flush=True)
    # the streams compute the same thing: outputs must agree (GroupNorm sums are fp64 atomics: last-bit differences only)
    ref = trk[0].out.float()
    worst = max(float((trk[j].out.float() - ref).abs().max() / ref.abs().max()) for j in range(1, kmax)) if kmax > 1 else 0.0
    print(json.dumps({"max_rel_diff_between_streams": worst}))
    if args.out:
        json.dump({"model": args.model, "size": [H, W], "rows": res, "max_rel_diff_between_streams": worst}, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
