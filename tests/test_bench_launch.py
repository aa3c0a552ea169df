"""`python bench.py --gpus N` must start its own ranks when no launcher did (north_star: "report 1/2/4/8-GPU throughput").
CPU-only check of that path: --launch-check makes the ranks rendezvous over gloo on 127.0.0.1 and skip the GPU work."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    return env


def test_self_launch_two_ranks_prints_one_json_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec == {"launch_check": True, "world": 2, "max_rank_seen": 1}


def test_under_an_external_launcher_nothing_is_spawned():
    """With WORLD_SIZE set (torch.distributed.run did the launching) the process is a rank, not a parent."""
    env = dict(_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--launch-check"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])["world"] == 1


def test_failing_rank_fails_the_launch():
    """Without a GPU the real ranks cannot run: every child exits non-zero and so must the parent (no JSON line, no hang)."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without a GPU")
    env = dict(_env(), ODEHIP_BENCH_REHEARSAL="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
