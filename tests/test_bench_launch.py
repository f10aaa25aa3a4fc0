"""CPU suite: `python bench.py --gpus N` starts its N ranks itself (no torchrun), the ranks rendezvous and reduce over
torch.distributed (gloo here, RCCL on the GPU node), rank 0 prints one JSON line with n_gpus == N; a failing rank makes
the launcher exit non-zero instead of hanging.  (VERDICT r1 item 1a / ADVICE r1.)"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _run(*argv, env_extra=None, timeout=180):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv], capture_output=True, text=True, env=env, timeout=timeout)


def test_bench_starts_its_own_ranks():
    r = _run("--gpus", "2", "--launch-check", env_extra={"GSM_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and d["backend"] == "gloo" and d["max_rank_seen"] == 1.0


def test_gpus_flag_must_match_the_launcher_world():
    r = _run("--gpus", "4", "--launch-check", env_extra={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr


def test_failed_rank_fails_the_launch():
    """Without a GPU every rank fails when it creates its engine (no CPU fallback): the launcher must report that."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a machine without a GPU")
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", env_extra={"GSM_DIST_BACKEND": "gloo"})
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
