""""persistent walk, then collective, then persistent walk" in ONE process on one GPU.  The persistent kernels
hold every CU and their workgroups spin-wait for each other under an ordinary launch; RCCL brings its own kernels, streams and
device allocations into the same process.  Every other multi-rank test switches the walk off (two ranks share one card there),
so this is the one place where both run together: one rank, persistent path ON, a real process group (RCCL when it comes up on
the box, gloo otherwise is a separate case), three training steps with the gradient all-reduce between them."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def _run(backend, tmp_path):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if not k.startswith("ODEHIP_PERSISTENT")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = tmp_path / f"{backend}.json"
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_persist_collective_worker.py")
    r = subprocess.run([sys.executable, worker, backend, str(out)], env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    return json.load(open(out))


@pytest.mark.parametrize("backend", ["nccl", "gloo"])
def test_training_steps_on_the_persistent_walk_with_a_collective_between_them(cuda, tmp_path, backend):
    if os.environ.get("ODEHIP_PERSISTENT") == "0":
        pytest.skip("persistent path switched off for this run")
    rec = _run(backend, tmp_path)
    assert rec["backend"] == backend and rec["world"] == 1
    # every training step = one persistent saving forward + one persistent reverse sweep
    assert rec["persistent_launches"] == 8, rec
    assert rec["persistent_error"] == 0 and rec["finite"] and rec["x_ok"]
    assert rec["identical_across_steps"] and rec["allreduce_world1_is_identity"]
    assert rec["bucket_elems"] == 5 * (64 * 64 * 9 + 64)
