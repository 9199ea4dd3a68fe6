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
