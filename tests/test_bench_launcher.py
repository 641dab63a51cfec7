"""CPU: `python bench.py --gpus N` must start N ranks by itself (the driver's SCALE command has no torchrun around
it; the reference's launch is one command too: train.sh:1, train.py:98-100) and must never fall back to fewer ranks."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True,
                          env=e, timeout=300)


def test_self_launch_two_ranks_gloo_dry_run():
    r = _run("--gpus", "2", "--backend", "gloo", "--dry-run", "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0's line, relayed once
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["process_group_ranks"] == 2
    assert line["steps"] == 2 and line["warmup"] == 1


def test_refuses_more_ranks_than_gpus():
    # this container has no GPU: a real (non-dry) 2-rank run must fail loudly, not run 1 rank
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("2 GPUs visible")
    r = _run("--gpus", "2")
    assert r.returncode != 0
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())
    assert "refusing" in r.stderr


def test_gpus_flag_must_match_world_size():
    r = _run("--gpus", "2", "--backend", "gloo", "--dry-run",
             env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "disagrees" in (r.stderr + r.stdout)


def test_failed_rank_fails_the_launch():
    # gloo without --dry-run is refused inside every child: the parent must report failure and print no line
    r = _run("--gpus", "2", "--backend", "gloo", env={"VCVITS_BENCH_SKIP_DEVICE_CHECK": "1"})
    assert r.returncode != 0
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())
