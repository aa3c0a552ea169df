"""Diagnostic: per-layer timeline of logical workgroup 0 inside the persistent trajectory launch (conv_wino.hip).
Stamps (100 MHz): 0 layer start (producer), 1 partners' previous layer seen, 2 first input chunk + weights landed, 3 first input
transform done, 4 consumers past the first barrier, 5 last MFMA issued, 6 stores issued, 7 stores acknowledged.
  python tools/persist_stamps.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ode_rl_amd  # noqa: E402

dev = torch.device("cuda:0")
lib = ode_rl_amd._lib.load()
torch.manual_seed(0)
f = ode_rl_amd.ODEFunc(64, 64, 3, 64, False, "relu", final_act=False).to(dev)
z0 = torch.randn(64, 64, 16, 16, device=dev) * 0.5
t = torch.arange(10, 20, dtype=torch.float64, device=dev) / 20
buf = torch.zeros(64 * 8, dtype=torch.int64, device=dev)
with torch.no_grad():
    for _ in range(3):
        ode_rl_amd.odeint(f, z0, t, method="rk4")
    lib.odehip_set_debug_buffer(buf.data_ptr())
    ode_rl_amd.odeint(f, z0, t, method="rk4")
    torch.cuda.synchronize()
    lib.odehip_set_debug_buffer(None)
s = buf.cpu().view(64, 8).numpy().astype("int64")
print("layer   wait  load  xform  ->bar | mfma  epil  ack | layer total (us)")
for l in range(20, 45):
    r = s[l] - s[l][0]
    nxt = s[l + 1][0] - s[l][0]
    print(f"{l:5d} {r[1] / 100:6.2f} {(r[2] - r[1]) / 100:5.2f} {(r[3] - r[2]) / 100:6.2f} {(r[4] - r[3]) / 100:6.2f} | {(r[5] - r[4]) / 100:5.2f} "
          f"{(r[6] - r[5]) / 100:5.2f} {(r[7] - r[6]) / 100:5.2f} | {nxt / 100:6.2f}   (prod. start -> cons. ack {r[7] / 100:6.2f})")
