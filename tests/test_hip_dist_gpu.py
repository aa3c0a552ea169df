"""N > 1 on the GPU path: two ranks (sharing the one GPU of the test box, gloo collectives) integrate the two halves of a
batch with dopri5 under exact-global step control and must reproduce the single-process run on the
full batch: same accepted step sequence, trajectories and gradients to round-off (only the summation order of the error
norm differs)."""
import os
import subprocess
import sys

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def test_two_ranks_exact_global_dopri5_match_single_process(cuda, tmp_path):
    import ode_rl_amd
    torch.manual_seed(0)
    f = ode_rl_amd.ODEFunc(64, 64, 3, 64, False, "relu", final_act=False)
    with torch.no_grad():
        for i in (0, 2, 4, 6):
            f.gradient_net[i].weight.mul_(0.15)
            f.gradient_net[i].bias.copy_(torch.where(torch.arange(64) % 2 == 0, 2.5, -2.5))
        f.gradient_net[8].weight.mul_(4.0)
    g = torch.Generator().manual_seed(3)
    z0 = torch.randn(4, 64, 16, 16, generator=g) * 0.5
    z0[2:] *= 3.0                                   # the two shards have different error norms: per-shard control would differ
    t = torch.tensor([0.0, 0.3, 0.5, 1.0], dtype=torch.float64)
    gout = torch.randn(4, 4, 64, 16, 16, generator=g)
    blob = {"state": f.state_dict(), "z0": z0, "gout": gout, "t": t, "rtol": 1e-4, "atol": 1e-5}
    torch.save(blob, tmp_path / "in.pt")

    import socket
    with socket.socket() as sk:   # a free rendezvous port on the loopback
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    # two ranks share ONE card here: the persistent launches (every CU, workgroups waiting for each other) assume one process per
    # GPU, so the ranks use one launch per layer -- the results are bit-identical either way
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", ODEHIP_PERSISTENT="0")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dist_gpu_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(tmp_path / f"out{r}.pt"), str(tmp_path / "in.pt")],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0

    fd = f.to(cuda)
    zd = z0.to(cuda).requires_grad_(True)
    sol = ode_rl_amd.odeint(fd, zd, t, rtol=1e-4, atol=1e-5, method="dopri5")
    full = dict(ode_rl_amd.last_stats)
    sol.backward(gout.to(cuda))
    outs = [torch.load(tmp_path / f"out{r}.pt") for r in range(2)]
    for o in outs:
        assert (o["n_accept"], o["n_reject"]) == (full["n_accept"], full["n_reject"])
        assert all(abs(a[1] - b[1]) <= 1e-5 * abs(b[1]) for a, b in zip(o["accepted"], full["accepted"]))
    assert rel_l2(torch.cat([o["sol"] for o in outs], 1), sol.detach()) <= 1e-5
    assert rel_l2(torch.cat([o["gz"] for o in outs], 0), zd.grad) <= 1e-4
    for k, p in enumerate(fd.parameters()):
        assert rel_l2(outs[0]["gp"][k], p.grad) <= 1e-4      # summed over ranks == full-batch gradient
        assert torch.equal(outs[0]["gp"][k], outs[1]["gp"][k])
    # and per-shard control really would have differed: shard 0 alone takes a different step sequence
    with torch.no_grad():
        ode_rl_amd.odeint(fd, z0[:2].to(cuda), t, rtol=1e-4, atol=1e-5, method="dopri5")
    alone = ode_rl_amd.last_stats["accepted"]
    assert len(alone) != len(full["accepted"]) or any(abs(a[1] - b[1]) > 1e-3 * abs(b[1]) for a, b in zip(alone, full["accepted"]))
