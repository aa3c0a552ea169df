"""Diagnostic: per-row timeline of logical workgroup 0 inside the LAST attempted step (program P2: 2 elementwise rows + 60 conv rows)
of the device-driven adaptive adjoint (adjoint_device.hip) -- stamps as in tools/persist_stamps.py.
  python tools/adjoint_stamps.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ode_rl_amd  # noqa: E402

dev = torch.device("cuda:0")
lib = ode_rl_amd._lib.load()
torch.manual_seed(0)
f = ode_rl_amd.ODEFunc(64, 64, 3, 64, False, "relu", final_act=False).to(dev)
z0 = (torch.randn(64, 64, 16, 16, device=dev) * 0.5).requires_grad_(True)
t = torch.arange(10, 20, dtype=torch.float64) / 20
gout = torch.randn(10, 64, 64, 16, 16, device=dev)
buf = torch.zeros(64 * 8, dtype=torch.int64, device=dev)


def step():
    f.zero_grad()
    o = ode_rl_amd.odeint_adjoint(f, z0, t, rtol=1e-5, atol=1e-6, method="dopri5", adjoint_options={"norm": "seminorm"})
    o.backward(gout)


for _ in range(2):
    step()
torch.cuda.synchronize()
lib.odehip_set_debug_buffer(buf.data_ptr())
step()
torch.cuda.synchronize()
lib.odehip_set_debug_buffer(None)
s = buf.cpu().view(64, 8).numpy().astype("int64")
print("row    wait  load  xform  ->bar | mfma  epil  ack | row total (us)")
tot = []
for l in range(2, 61):
    if s[l][0] == 0 or s[l + 1][0] == 0:
        continue
    r = s[l] - s[l][0]
    nxt = s[l + 1][0] - s[l][0]
    tot.append(nxt / 100)
    kind = "f" if ((l - 2) % 10) < 5 else "d"
    print(f"{l:3d}{kind}{(l - 2) % 5} {r[1] / 100:6.2f} {(r[2] - r[1]) / 100:5.2f} {(r[3] - r[2]) / 100:6.2f} {(r[4] - r[3]) / 100:6.2f} | {(r[5] - r[4]) / 100:5.2f} "
          f"{(r[6] - r[5]) / 100:5.2f} {(r[7] - r[6]) / 100:5.2f} | {nxt / 100:6.2f}")
if tot:
    print(f"mean row {sum(tot) / len(tot):.2f} us over {len(tot)} rows")
