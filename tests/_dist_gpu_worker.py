"""Worker of tests/test_hip_dist_gpu.py: one rank of a 2-rank job that shares ONE GPU (gloo for the collectives).
Integrates its shard of the batch with dopri5 under exact-global step control, then backward; writes results to a file."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    out_path = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    import ode_rl_amd
    from ode_rl_amd import dist as odist
    dev = torch.device("cuda", 0)
    blob = torch.load(sys.argv[2])
    f = ode_rl_amd.ODEFunc(64, 64, 3, 64, False, "relu", final_act=False)
    f.load_state_dict(blob["state"])
    f = f.to(dev)
    z0 = odist.shard_batch(blob["z0"], rank, world).to(dev).requires_grad_(True)
    gout = odist.shard_batch(blob["gout"], rank, world, dim=1).to(dev)
    t = blob["t"]
    odist.enable_global_step_control(dev)
    sol = ode_rl_amd.odeint(f, z0, t, rtol=blob["rtol"], atol=blob["atol"], method="dopri5")
    stats = dict(ode_rl_amd.last_stats)
    sol.backward(gout)
    odist.allreduce_gradients(f.parameters(), average=False)
    odist.disable_global_step_control()
    torch.save({"sol": sol.detach().cpu(), "gz": z0.grad.cpu(), "gp": [p.grad.cpu() for p in f.parameters()],
                "n_accept": stats["n_accept"], "n_reject": stats["n_reject"], "accepted": stats["accepted"]}, out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
