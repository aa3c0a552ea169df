"""Soak of round 3's persistent launches: the adaptive walk (dopri5 attempts, the saving forward, the one-walk backward through dopri5,
the device-driven adaptive adjoint with its controller ticks) and the sixteen-workgroups-per-sample walk (B <= 16: forward, saving
forward, reverse sweep).  Every result must equal the first one bit for bit, the step counts must not move, and no capped wait may
give up.   python tools/soak_adaptive.py [reps]"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ode_rl_amd  # noqa: E402

lib = ode_rl_amd._lib.load()
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
torch.manual_seed(0)
f = ode_rl_amd.ODEFunc(64, 64, 3, 64, False, "relu", final_act=False).to(dev)
p0 = lib.odehip_persistent_trajectory_launches()
t_start = time.time()


def same(a, b):
    return all(torch.equal(x, y) for x, y in zip(a, b))


def grads(z):
    return [z.grad.clone()] + [p.grad.clone() for p in f.parameters()]


for B, T in ((64, 10), (8, 6), (70, 5), (128, 4)):
    z0 = (torch.randn(B, 64, 16, 16, device=dev) * 0.5).requires_grad_(True)
    t = torch.arange(T, 2 * T, dtype=torch.float64) / (2 * T)
    gout = torch.randn(T, B, 64, 16, 16, device=dev)
    first = {}
    for i in range(reps):
        for name, fn in (("dopri5 fwd+bwd", lambda: ode_rl_amd.odeint(f, z0, t, rtol=1e-4, atol=1e-5, method="dopri5")),
                         ("dopri5 adjoint", lambda: ode_rl_amd.odeint_adjoint(f, z0, t, rtol=1e-5, atol=1e-6, method="dopri5",
                                                                              adjoint_options={"norm": "seminorm"}))):
            if name == "dopri5 adjoint" and (i % 4 or B > 64):
                continue
            f.zero_grad()
            z0.grad = None
            out = fn()
            st = dict(ode_rl_amd.last_stats)
            out.backward(gout)
            adj = dict(ode_rl_amd.last_adjoint_stats) if name == "dopri5 adjoint" else {}
            cur = [out.detach().clone()] + grads(z0)
            key = (name,)
            if key not in first:
                first[key] = (cur, st.get("nfe"), adj.get("nfe"))
            else:
                assert same(cur, first[key][0]) and st.get("nfe") == first[key][1] and adj.get("nfe") == first[key][2], (B, name, i)
    torch.cuda.synchronize()
    print(f"B {B} T {T}: {reps} dopri5 training steps (saving forward + one-walk backward), {len(range(0, reps, 4)) if B <= 64 else 0} adaptive-adjoint "
          f"training steps: identical; error word {lib.odehip_persistent_error(0)}; {time.time() - t_start:.0f} s", flush=True)

for B in (1, 4, 9, 16):   # the 16-workgroup walk
    z0 = (torch.randn(B, 64, 16, 16, device=dev) * 0.5).requires_grad_(True)
    t = torch.arange(10, 20, dtype=torch.float64) / 20
    gout = torch.randn(10, B, 64, 16, 16, device=dev)
    ref = None
    for i in range(reps * 3):
        f.zero_grad()
        z0.grad = None
        out = ode_rl_amd.odeint(f, z0, t, method="rk4")
        out.backward(gout)
        with torch.no_grad():
            inf = ode_rl_amd.odeint(f, z0.detach(), t, method="rk4")
        cur = [out.detach().clone(), inf] + grads(z0)
        if ref is None:
            ref = cur
        else:
            assert same(cur, ref), (B, i)
    torch.cuda.synchronize()
    print(f"B {B}: {reps * 3} rk4 training steps + forward trajectories on the 16-workgroup walk: identical; error word "
          f"{lib.odehip_persistent_error(0)}; {time.time() - t_start:.0f} s", flush=True)
print("persistent launches counted:", lib.odehip_persistent_trajectory_launches() - p0)
