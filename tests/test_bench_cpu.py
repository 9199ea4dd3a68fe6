"""bench.py plumbing that needs no GPU: `--gpus N` outside a torchrun environment must launch N ranks itself
(VERDICT r01 missing #1; reference strategy: external/lib/test/evaluation/running.py:111-120, one process per GPU)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--launch-check", *extra], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout          # rank 0 prints ONE json line
    # ... and NOTHING else on stdout (gloo's "[Gloo] Rank r is connected to ..." chatter is redirected to stderr while the groups form)
    assert [l for l in out.stdout.splitlines() if l.strip()] == lines, out.stdout
    return json.loads(lines[0])


def test_gpus_2_self_launches_two_ranks():
    r = _run("--gpus", "2")
    assert r["n_gpus"] == 2 and r["backend"] in ("gloo", "nccl")


def test_gpus_1_stays_single_process():
    r = _run("--gpus", "1")
    assert r["n_gpus"] == 1 and r["backend"] == "none"


def test_mix_task_assignment_and_gathers_world4():
    """BASELINE.json configs[4] (MOT+SOT mix): --task mix gives the first half of the ranks the MOT loop and the second half the SOT
    step; the in-run row gather (ragged per rank) and the byte-string gather run on every rank (gloo here, RCCL on GPUs)."""
    r = _run("--gpus", "4", "--task", "mix")
    assert r["n_gpus"] == 4
    assert r["tasks"] == ["mot", "mot", "sot", "sot"]
    assert r["gathered_rows"] == sum(3 + k for k in range(4))
    assert r["gathered_strings"] == sum(1 + k % 2 for k in range(4))


def test_mix_single_rank_is_sot():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.rank_task("mix", "unicorn_track_large", 0, 1) == ("sot", "unicorn_track_large")
    assert bench.rank_task("mix", "unicorn_track_large", 0, 2) == ("mot", "unicorn_track_large_mot_challenge")
    assert bench.rank_task("mix", "unicorn_track_large", 1, 2) == ("sot", "unicorn_track_large")
    assert bench.rank_task("mot", "unicorn_track_large", 1, 2) == ("mot", "unicorn_track_large")


def _run_env(env_extra, *extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--launch-check", *extra], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_rccl_probe_child_processes_run_and_report():
    """init_dist(): the first RCCL contact happens in one child process per rank (own rendezvous port, timeout, exact-PID kill); here
    the children rendezvous over gloo (no GPU) and every rank must report a good probe."""
    r = _run_env({"UNI_BENCH_PROBE": "1"}, "--gpus", "2")
    assert r["dist"]["rccl_probe"]["ok_all_ranks"] is True and r["dist"]["control"] == "gloo"
    assert r["n_gpus"] == 2 and r["gathered_rows"] == 3 + 4


def test_failed_rccl_probe_degrades_to_the_gloo_gather():
    """a probe that dies on every rank must not lose the run: the gathers go over gloo and the line says so"""
    r = _run_env({"UNI_BENCH_PROBE": "1", "UNI_BENCH_PROBE_FAIL": "1"}, "--gpus", "2")
    assert r["dist"]["rccl_probe"]["ok_all_ranks"] is False and r["dist"]["data"] == "gloo"
    assert r["n_gpus"] == 2 and r["gathered_rows"] == 3 + 4 and r["gathered_strings"] == 1 + 2


def test_timed_loop_keeps_calling_collectives_after_a_rank_error():
    """N > 1: an exception in a rank's step is recorded and the rank keeps taking part in every gather (with the rows it has), so
    the other ranks never hang; N = 1 (errs=None): the exception propagates."""
    sys.path.insert(0, ROOT)
    import pytest
    import bench

    class S:
        n = 0

        def step(self, i):
            S.n += 1
            if i == 3:
                raise RuntimeError("boom")

    calls = []
    errs = []
    bench.timed([S()], steps=6, warmup=2, barrier=lambda: None, gather=lambda st: calls.append(1) or 0, gather_every=2, errs=errs)
    assert len(calls) == 4 and len(errs) == 1 and "boom" in errs[0] and S.n == 4      # steps 0..3 ran, 4..7 skipped, all 4 gathers called
    with pytest.raises(RuntimeError):
        bench.timed([S()], steps=6, warmup=2, barrier=lambda: None)


def test_gpu_sensor_summary_uses_only_the_timed_window():
    """clock / power statistics come from the samples BETWEEN the two barriers of the timed steps (median, p10, p90,
    fraction of samples within 3 % of the power cap), not from a mean over warm-up + ramp; timed() hands the window out."""
    sys.path.insert(0, ROOT)
    import bench
    s = bench.GpuSensor.__new__(bench.GpuSensor)
    s.paths, s.period = ("/nonexistent", "/nonexistent", "/nonexistent"), 0.005
    # warm-up / ramp: low clock, low power (t < 10); timed window: 2000 MHz at 1380 W of a 1400 W cap, one dip
    s.samples = [(900.0, 300.0, float(t)) for t in range(10)] + [(2000.0, 1380.0, 10.0 + t) for t in range(19)] + [(1500.0, 1000.0, 29.0)]
    whole, win = s.summary(), s.summary(window=(10.0, 29.0))
    assert whole["samples"] == 30 and win["samples"] == 20
    assert win["sclk_mhz_median"] == 2000.0 and win["sclk_mhz_p10"] == 2000.0 and win["sclk_mhz_min"] == 1500.0
    assert win["power_w_median"] == 1380.0 and win["power_cap_w"] is None and win["frac_samples_within_3pct_of_power_cap"] is None
    assert whole["sclk_mhz_mean"] < 1700 < win["sclk_mhz_mean"]
    assert s.summary(window=(100.0, 200.0)) is None

    class _S:
        def step(self, i):
            pass
    w = []
    dt, own, _ = bench.timed([_S()], 3, 1, lambda: None, window=w)
    assert len(w) == 2 and abs((w[1] - w[0]) - dt) < 1e-6
