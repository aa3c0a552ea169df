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
    coll = rec.pop("collective")
    assert rec == {"launch_check": True, "world": 2, "max_rank_seen": 1}
    # what the process group itself says: backend, its own world size, one identity per rank, all distinct
    assert coll == {"backend": "gloo", "world_size": 2, "ranks": [0, 1], "distinct_pids": 2}


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


def test_launcher_parent_never_loads_torch_or_a_gpu_runtime():
    """the parent of a --gpus N launch must not touch torch.cuda (on ROCm device_count() can fall through to
    hipGetDeviceCount, i.e. open /dev/kfd, in a process that then spawns).  It does not even import torch: the spawn loop runs
    with Popen replaced by a recorder and `torch` / any amdhip / hsa library must be absent from the process afterwards."""
    code = r"""
import os, sys, types
sys.argv = ["bench.py", "--gpus", "3", "--steps", "1"]
sys.path.insert(0, %r)
import subprocess
spawned = []
class FakeProc:
    def __init__(self, argv, env=None, stdout=None):
        spawned.append((argv, env))
    def poll(self):
        return 0
subprocess.Popen = FakeProc
os.environ["HIP_VISIBLE_DEVICES"] = "0,1,2"       # the count comes from the environment / sysfs, never from HIP
os.environ.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
import bench
assert bench.visible_gpu_count() == 3
try:
    bench.main()
except SystemExit as e:
    assert e.code == 0, e.code
assert len(spawned) == 3
assert [e["RANK"] for _, e in spawned] == ["0", "1", "2"] and all(e["WORLD_SIZE"] == "3" and e["MASTER_ADDR"] == "127.0.0.1" for _, e in spawned)
assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for _, e in spawned)
assert "torch" not in sys.modules, "launcher parent imported torch"
maps = open("/proc/self/maps").read()
assert "libamdhip64" not in maps and "libhsa-runtime" not in maps, "launcher parent mapped a GPU runtime"
os.environ["HIP_VISIBLE_DEVICES"] = "0"
try:
    bench.main()
    raise AssertionError("3 ranks on 1 visible GPU must be refused")
except SystemExit as e:
    assert "only 1 GPU" in str(e.code)
print("ok")
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr[-3000:]


def test_parent_kills_a_rank_that_ignores_sigterm():
    """terminate() alone lets a rank stuck in an uninterruptible wait hang the parent for ever."""
    code = r"""
import os, sys, time, signal
sys.path.insert(0, %r)
import subprocess, types
real = subprocess.Popen
def fake(argv, env=None, stdout=None):
    if env["RANK"] == "0":
        return real([sys.executable, "-c", "import sys; sys.exit(7)"])
    return real([sys.executable, "-c", "import signal, time; signal.signal(signal.SIGTERM, signal.SIG_IGN); time.sleep(600)"])
subprocess.Popen = fake
import bench
a = types.SimpleNamespace(gpus=2, launch_check=True)
t0 = time.time()
try:
    bench.self_launch(a, grace_s=1.0)
except SystemExit as e:
    assert e.code == 7, e.code
assert time.time() - t0 < 30
print("ok")
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr[-3000:]
