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
