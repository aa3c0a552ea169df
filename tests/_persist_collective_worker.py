"""Worker of tests/test_hip_persist_collective.py: ONE rank on ONE GPU with the persistent walk ON and a real process group in
the same process: training step (two persistent launches + weight gradients) -> the step's ONE gradient all-reduce -> training
step again.  argv: backend ("nccl" = RCCL, or "gloo"), output file."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    backend, out_path = sys.argv[1], sys.argv[2]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
    import ode_rl_amd
    from ode_rl_amd import dist as odist
    lib = ode_rl_amd._lib.load()
    torch.manual_seed(0)
    f = ode_rl_amd.ODEFunc(64, 64, 3, 64, False, "relu", final_act=False).to(dev)
    z0 = (torch.randn(64, 64, 16, 16, generator=torch.Generator().manual_seed(5)) * 0.5).to(dev)
    gout = torch.randn(10, 64, 64, 16, 16, generator=torch.Generator().manual_seed(6)).to(dev)
    t = torch.arange(10, 20, dtype=torch.float64) / 20

    def train_step():
        zz = z0.detach().requires_grad_(True)
        f.zero_grad(set_to_none=False)
        sol = ode_rl_amd.odeint(f, zz, t, method="rk4")
        sol.backward(gout)
        grads_before = [p.grad.clone() for p in f.parameters()]
        n = odist.allreduce_gradients(f.parameters())          # world 1: sum / 1 -- must hand back the same bits
        return sol.detach().clone(), zz.grad.clone(), grads_before, [p.grad.clone() for p in f.parameters()], n

    p0 = lib.odehip_persistent_trajectory_launches()
    runs = [train_step() for _ in range(3)]                     # walk, collective, walk, collective, walk
    x = torch.ones(1 << 20, device=dev)
    dist.all_reduce(x)                                          # a bandwidth-sized collective between two walks as well
    runs.append(train_step())
    torch.cuda.synchronize()
    launches = lib.odehip_persistent_trajectory_launches() - p0
    err = int(lib.odehip_persistent_error(0))
    same = all(torch.equal(a[0], runs[0][0]) and torch.equal(a[1], runs[0][1]) and all(torch.equal(u, v) for u, v in zip(a[3], runs[0][3]))
               for a in runs[1:])
    reduced_same = all(torch.equal(u, v) for u, v in zip(runs[0][2], runs[0][3]))
    rec = {"backend": dist.get_backend(), "world": dist.get_world_size(), "persistent_launches": int(launches), "persistent_error": int(err),
           "identical_across_steps": bool(same), "allreduce_world1_is_identity": bool(reduced_same), "bucket_elems": int(runs[0][4]),
           "finite": bool(torch.isfinite(runs[0][0]).all()) and all(bool(torch.isfinite(g).all()) for g in runs[0][3]),
           "x_ok": bool((x == 1).all())}
    json.dump(rec, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
