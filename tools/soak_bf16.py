"""Soak of the one-launch bf16 paths (whole-trajectory forward, saving forward, segmented reverse sweep with the weight gradients on
the side stream): the same training step N times -- every output must be BITWISE identical to the first iteration's (the design has no
float atomics and a fixed summation order), the persistent error word must stay clear.
  python tools/soak_bf16.py [--iters 300] [--batch 128] [--frames 40]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--iters", type=int, default=300)
    p.add_argument("--batch", type=int, default=128)
    p.add_argument("--frames", type=int, default=40)
    a = p.parse_args()
    import ode_rl_amd
    ode_rl_amd.set_compute_dtype("bf16")
    lib = ode_rl_amd._lib.load()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    f = ode_rl_amd.ODEFunc(64, 64, 3, 64, False, "relu", final_act=False).to(dev)
    T = a.frames
    z0 = torch.randn(a.batch, 64, 16, 16, device=dev) * 0.5
    t = torch.arange(T, 2 * T, dtype=torch.float64) / (2 * T)
    gout = torch.randn(T, a.batch, 64, 16, 16, device=dev)
    ref = None
    other = torch.zeros(64 << 20, device=dev)   # unrelated allocator traffic between the steps
    for it in range(a.iters):
        f.zero_grad()
        z = z0.clone().requires_grad_(True)
        out = ode_rl_amd.odeint(f, z, t, method="rk4")
        out.backward(gout)
        got = [out.detach(), z.grad] + [p_.grad for p_ in f.parameters()]
        if it % 3 == 0:
            other.add_(1.0)
            with torch.no_grad():
                inf = ode_rl_amd.odeint(f, z0, t, method="rk4")
            assert torch.equal(inf, out.detach()), f"iteration {it}: inference and training forward differ"
        if ref is None:
            ref = [g.clone() for g in got]
            assert all(bool(torch.isfinite(g).all()) for g in ref)
        else:
            for k, (g, r) in enumerate(zip(got, ref)):
                assert torch.equal(g, r), f"iteration {it}: output {k} differs from the first iteration"
        assert lib.odehip_persistent_error(0) == 0
    torch.cuda.synchronize()
    print(f"soak ok: {a.iters} identical bf16 training steps (B={a.batch}, T={T}), persistent launches {lib.odehip_persistent_trajectory_launches()}")


if __name__ == "__main__":
    main()
