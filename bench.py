#!/usr/bin/env python
"""bench.py — frames/s of Unicorn's per-frame step on MI355X (BASELINE.json metric).

A "step" = one frame through the hot path (SURVEY.md §8d unit of work): ConvNeXt+PAFPN on the current
frame, ref<->cur deformable interaction, 2x embedding upsample, dense HWxHW correlation + prior
propagation (fp32), prior pyramid, unified head.  The reference-frame backbone is cached (computed once,
external/lib/test/tracker/unicorn_sot.py:49) and is outside the timed region, as are H2D copies: frames are
resident in HBM before timing starts.  Weights: synthetic (oracle/synth.py), data: synthetic clip.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model unicorn_track_large] [--task sot]

N>1: launched by torch.distributed.run, one process per GPU, one independent video stream per rank
(streams shard one-per-GPU, no data-path collective; RCCL all_gather only for the result rows).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="unicorn_track_large")
    ap.add_argument("--task", default="sot", choices=["sot", "mot"])
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--corr-precision", type=int, default=1, choices=[0, 1],
                    help="0 = fp32 MFMA correlation, 1 = fp32-equivalent bf16x3 split (default)")
    ap.add_argument("--cpu-frames", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16, help="frames of the stream processed per step (time-batched: the SOT step of a frame\n                    depends only on the cached reference frame, unicorn_sot.py:78-108, so consecutive frames are independent)")
    ap.add_argument("--streams-per-gpu", type=int, default=1, help="independent video streams multiplexed on one GPU (own HIP stream + context each)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)   # "nccl" == RCCL on ROCm
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback for the measured path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import synth
    import unicorn_oracle as uo
    from unicorn_amd import _lib as L
    from unicorn_amd.models import Unicorn
    from unicorn_amd.ops import corr_softmax_pv, label_map_s8, prior_pyramid, sample_embeddings

    H, W = args.height, args.width
    CORR_PREC = args.corr_precision
    cfg = uo.CONFIGS[args.model]
    P = synth.synth_state_dict(cfg)
    S = max(1, args.streams_per_gpu)
    models = []
    for _ in range(S):
        mdl = Unicorn(args.model).cuda(local_rank)
        mdl.load_state_dict(P)
        models.append(mdl)
    model = models[0]
    hip_streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(S - 1)]
    # one independent synthetic stream per (rank, slot), frames resident in HBM
    n_frames = 4
    clips = []
    for j in range(S):
        fr, bx = synth.synth_clip(H, W, n_frames + 1, seed=1 + rank * S + j)
        clips.append(([f.to(dev) for f in fr], bx))
    frames, box = clips[0]
    state = []
    with torch.no_grad():
        for j in range(S):
            _, dp = models[j](imgs=clips[j][0][0], mode="backbone")          # reference frame: once, untimed
            state.append((dp, label_map_s8(clips[j][1], H, W, dev)))
    torch.cuda.synchronize()
    d_pre, lbs = state[0]
    results = torch.zeros((args.steps + args.warmup, 8), device=dev)

    def step(i):
        if S == 1:
            return step_one(i, 0)
        j = i % S
        with torch.cuda.stream(hip_streams[j]):
            step_one(i, j)

    NB = max(1, args.batch)
    batches = [[torch.cat([clips[j][0][1 + (k + t) % n_frames] for t in range(NB)], 0) for k in range(n_frames)] for j in range(S)]

    def step_one(i, j):
        model = models[j]
        d_pre, lbs = state[j]
        img = batches[j][(i // S) % n_frames]                    # (NB,3,H,W): NB consecutive frames of the stream
        with torch.no_grad():
            if args.task == "sot":
                fpn, d_cur = model(imgs=img, mode="backbone")
                f_pre, f_cur = model(seq_dict0=d_pre, seq_dict1=d_cur, mode="interaction")
                e_pre = model(feat=f_pre, mode="upsample")
                e_cur = model(feat=f_cur, mode="upsample")
                pred = torch.cat([corr_softmax_pv(e_pre[b].flatten(-2), e_cur[b].flatten(-2), lbs, precision=CORR_PREC) for b in range(NB)], 0)
                pri = prior_pyramid(pred.view(1, NB, d_cur["h"] * 2, d_cur["w"] * 2))
                pri = tuple(t.transpose(0, 1).contiguous() for t in pri)
                out = model.head(fpn, pri, mode="sot")
                out = out[0] if cfg.mask else out
                # result rows = best-scoring anchor per frame (stand-in for NMS top-1; stays on device, no sync)
                best = torch.argmax(out[:, :, 4] * out[:, :, 5], 1)
                results[i, :6] = out[0, best[0], :6]
            else:   # evaluate_omni-style MOT step (mot_evaluator.py:991-1034)
                out, d_cur = model(img)
                out = out[0] if cfg.mask else out
                f_pre, f_cur = model(seq_dict0=d_pre, seq_dict1=d_cur, mode="interaction")
                e_cur = model(feat=f_cur, mode="upsample")
                for bi in range(NB):
                    sc = out[bi, :, 4] * out[bi, :, 5:].max(1)[0]
                    top = torch.topk(sc, 64)[1]
                    b = out[bi, top, :4]
                    boxes = torch.stack([b[:, 0] - b[:, 2] / 2, b[:, 1] - b[:, 3] / 2, b[:, 0] + b[:, 2] / 2, b[:, 1] + b[:, 3] / 2], 1)
                    emb = sample_embeddings(e_cur[bi:bi + 1], boxes)
                    results[i, :4] = boxes[0]
                    results[i, 4] = emb.sum()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # result gather (fixed-stride rows over RCCL), outside the timed region like the reference's end-of-eval gather
        from unicorn_amd.parallel import gather_result_rows
        rows = results.clone()
        rows[:, 0] = rank
        rows[:, 1] = torch.arange(rows.shape[0], device=dev)
        table = gather_result_rows(rows)
        assert table.shape[0] == world * rows.shape[0]
    fps = world * args.steps * NB / dt

    # ---------------- roofline leg: per-kernel-class HIP-event timing on the launch stream (rank 0) ----------------
    roof = None
    extra = {}
    if rank == 0:
        import ctypes as C
        prof_steps = 3
        buf = (C.c_double * 16)()
        L.check(L.lib().uni_prof_begin(model._ctx), "prof_begin")
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for i in range(prof_steps):
            step(i)
        L.check(L.lib().uni_prof_end(model._ctx, buf), "prof_end")
        v = list(buf)
        names = ["gemm", "dwconv7_ln", "gn_apply", "layernorm", "misc"]
        pf = prof_steps * NB     # per frame
        cls = {n: dict(ms=v[3 * i] / pf, work=v[3 * i + 1] / pf, launches=v[3 * i + 2] / prof_steps)
               for i, n in enumerate(names)}
        g = cls["gemm"]
        g["bytes"] = v[15] / prof_steps
        g["ms_step"], g["work_step"] = g["ms"] * NB, g["work"] * NB
        peak = 2500.0   # dense bf16 MFMA TFLOP/s (MI355X_MICROARCH.md)
        ach = g["work"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
        traffic, tsrc = None, None
        try:    # HBM-side bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic.py)
            tsrc = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("pmc_hbm_traffic.json"))[-1]
            t = json.load(open(os.path.join(ROOT, "profiles", tsrc)))["gemm_bf16_kernel"]
            traffic = round(t["fetch_bytes_per_launch"] + t["write_bytes_per_launch"])
        except Exception:
            pass
        roof = {"kernel": "gemm_bf16_kernel (all instantiations)", "bound": "mfma", "achieved": round(ach, 2), "peak": peak,
                "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": tsrc,
                "algorithmic_bytes_per_launch": round(g["bytes"] / max(g["launches"], 1)) if "bytes" in g else None,
                "avg_launch_us": round(1e3 * g["ms_step"] / max(g["launches"], 1), 2), "launches_per_step": g["launches"],
                "flops_per_frame": g["work"], "flops_per_launch": round(g["work_step"] / max(g["launches"], 1))}
        for n in ("dwconv7_ln", "gn_apply", "layernorm"):
            c_ = cls[n]
            extra[n] = {"ms_per_frame": round(c_["ms"], 4), "GBps": round(c_["work"] / (c_["ms"] * 1e-3) / 1e9, 1) if c_["ms"] > 0 else 0,
                        "launches": c_["launches"]}
        extra["gemm_ms_per_frame"] = round(g["ms"], 4)
        extra["misc_ms_per_frame"] = round(cls["misc"]["ms"], 4)
        if args.task == "sot":   # correlation kernel alone (torch events on the current stream == launch stream)
            with torch.no_grad():
                fpn, d_cur = model(imgs=frames[1], mode="backbone")
                f_pre, f_cur = model(seq_dict0=d_pre, seq_dict1=d_cur, mode="interaction")
                e_pre, e_cur = model(feat=f_pre, mode="upsample"), model(feat=f_cur, mode="upsample")
                a, b = e_pre[0].flatten(-2), e_cur[0].flatten(-2)
                corr_softmax_pv(a, b, lbs, precision=CORR_PREC)
                ev[0].record()
                for _ in range(5):
                    corr_softmax_pv(a, b, lbs, precision=CORR_PREC)
                ev[1].record()
                torch.cuda.synchronize()
            ms = ev[0].elapsed_time(ev[1]) / 5
            n = a.shape[1]
            # precision 1 issues 6 bf16 MFMA terms per fp32-equivalent product: effective peak = 2500 / 6 TFLOP/s
            extra["corr_fp32"] = {"ms": round(ms, 4), "TFLOPs": round(2.0 * n * n * 128 / (ms * 1e-3) / 1e12, 2),
                                  "peak_effective": 157.3 if CORR_PREC == 0 else round(2500.0 / 6, 1),
                                  "mode": "fp32 MFMA" if CORR_PREC == 0 else "bf16x3 split (6 exact partial products, fp32 accumulate)"}

    # ---------------- CPU baseline: the oracle (port of the reference) on the host cores, bounded sample ----------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cores = os.cpu_count() or 1
        cores = min(cores, 16)
        torch.set_num_threads(cores)
        cf, cbox = synth.synth_clip(H, W, 1 + args.cpu_frames, seed=1)
        with torch.no_grad():
            st = uo.sot_init(P, cfg, cf[0], cbox)
            t1 = time.perf_counter()
            for i in range(args.cpu_frames):
                if args.task == "sot":
                    uo.sot_step(P, cfg, st, cf[1 + i])
                else:
                    uo.mot_whole(P, cfg, cf[1 + i])
            cdt = time.perf_counter() - t1
        cpu = {"value": round(args.cpu_frames / cdt, 4), "unit": "frames/s", "cores": cores, "kind": "port",
               "sample": "%d frames of the same %s %s step at %dx%d, fp32, torch CPU (oracle/unicorn_oracle.py)"
                         % (args.cpu_frames, args.model, args.task, H, W)}

    if rank == 0:
        line = {
            "metric": "frames/sec @ 800x1280 %s" % args.model, "value": round(fps, 3), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4), "frames_per_step": NB, "ms_per_frame": round(1e3 * dt / (args.steps * NB), 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": "%s %s per-frame step %dx%d (backbone+FPN, deform interaction, embedding, fp32 correlation, "
                                   "head); one independent stream per GPU, %d consecutive frames per step" % (args.model, args.task.upper(), H, W, NB),
                       "model": args.model, "task": args.task, "frames_per_step": NB, "streams": world * S, "streams_per_gpu": S, "weights": "synthetic (oracle/synth.py)",
                       "corr_dtype": "f32" if CORR_PREC == 0 else "f32-equivalent (bf16x3 split operands, fp32 accumulate)", "accum": "f32"},
            "roofline": roof, "cpu_baseline": cpu, "kernels": extra,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
