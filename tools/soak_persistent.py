"""Soak of the persistent launches (fp32): repeated fixed-grid trajectories, dopri5 solves and training steps at several batch sizes
and both stack shapes; every result must equal the first one bit for bit and no capped wait may give up.
  python tools/soak_persistent.py [reps]"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ode_rl_amd  # noqa: E402

lib = ode_rl_amd._lib.load()
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
torch.manual_seed(0)
total = 0
for ch, layers in ((64, 3), (128, 2)):
    f = ode_rl_amd.ODEFunc(n_inputs=ch, n_outputs=ch, n_layers=layers, n_units=64, downsize=False, nonlinear="relu", final_act=False).to(dev)
    for B, T in ((64, 10), (4, 10), (70, 6), (128, 6)):
        z0 = torch.randn(B, ch, 16, 16, device=dev) * 0.5
        t = torch.arange(T, 2 * T, dtype=torch.float64, device=dev) / (2 * T)
        with torch.no_grad():
            ref = ode_rl_amd.odeint(f, z0, t, method="rk4")
            ref5 = ode_rl_amd.odeint(f, z0, t, method="dopri5")
            for i in range(reps):
                out = ode_rl_amd.odeint(f, z0, t, method="rk4")
                out5 = ode_rl_amd.odeint(f, z0, t, method="dopri5") if i % 10 == 0 else ref5
                if i % 50 == 49:
                    assert torch.equal(out, ref) and torch.equal(out5, ref5), (ch, B, i)
                total += 1
        zr = z0.clone().requires_grad_(True)
        grads = None
        for i in range(max(reps // 10, 3)):   # training steps: saving forward + reverse sweep + weight gradients
            for p in f.parameters():
                p.grad = None
            zr.grad = None
            ode_rl_amd.odeint(f, zr, t, method="rk4").square().sum().backward()
            g = [zr.grad.clone()] + [p.grad.clone() for p in f.parameters()]
            if grads is None:
                grads = g
            else:
                assert all(torch.equal(a, b) for a, b in zip(g, grads)), (ch, B, i)
        torch.cuda.synchronize()
        print(f"channels {ch} B {B} T {T}: {reps} trajectories, {reps // 10} dopri5 solves, {max(reps // 10, 3)} training steps identical; "
              f"persistent error word {lib.odehip_persistent_error(0)}", flush=True)
print("persistent launches counted:", lib.odehip_persistent_trajectory_launches())
