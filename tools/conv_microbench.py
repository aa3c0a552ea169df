"""Diagnostic: time back-to-back launches of the 3x3 64->64 conv (B=64) under ablation flags.
Launches are issued from C (odehip_debug_repeat_conv).  Run on the GPU box: python tools/conv_microbench.py [B]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ode_rl_amd  # noqa: E402,F401
from ode_rl_amd import hip_ops, _lib  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = hip_ops.nchw_to_q4(torch.randn(B, 64, 16, 16, device=dev))
w = hip_ops.pack_conv_weight(torch.randn(64, 64, 3, 3, device=dev) / 24)
bias = torch.randn(64, device=dev)
dst = torch.empty_like(x)
lib = _lib.load()
d = _lib.ConvDesc(src1=x.data_ptr(), src2=None, cin1=64, cin=64, cout=64, ks=3, batch=B, w_packed=w.data_ptr(),
                  bias=bias.data_ptr(), dst=dst.data_ptr(), relu=1)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
N = 500
names = {0: "full", 1: "no-dma", 2: "no-mfma", 3: "no-dma,no-mfma", 4: "no-epilogue", 5: "mfma only", 6: "dma only",
         7: "empty"}
for flags in (0, 1, 2, 3, 4, 5, 6, 7, 0):
    lib.odehip_set_debug_flags(flags)
    lib.odehip_debug_repeat_conv(ctypes.byref(d), 50, stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.odehip_debug_repeat_conv(ctypes.byref(d), N, stream)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / N
    print(f"flags={flags} ({names[flags]:>16}): {us:7.2f} us/launch   {2*B*64*64*9*256/us/1e6:7.1f} TFLOP/s-equivalent")
lib.odehip_set_debug_flags(0)

# ---- in-kernel stamps (diagnostic build path, flag 8): where does a workgroup spend its time?
dbg = torch.zeros(B * 4 * 8, dtype=torch.int64, device=dev)
lib.odehip_set_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
lib.odehip_set_debug_flags(8)
lib.odehip_debug_repeat_conv(ctypes.byref(d), 20, stream)
torch.cuda.synchronize()
lib.odehip_set_debug_flags(0)
lib.odehip_set_debug_buffer(None)
s = dbg.view(-1, 8).cpu().double()
t0 = s[:, 0].min()
rt = (s[:, :5] - t0) * 0.01  # us since the first workgroup started
print("per-workgroup stamps of the last launch (us since first WG start): median [min..max]")
for i, n in enumerate(["start", "dma issued", "chunk0 landed", "mfma done", "end"]):
    print(f"  {n:>14}: {rt[:, i].median():6.2f} [{rt[:, i].min():6.2f} .. {rt[:, i].max():6.2f}]")
cy = s[:, 5:8]
print("  shader cycles from start: dma issued %.0f, chunk0 landed %.0f, mfma done %.0f (median)" %
      tuple(cy.median(dim=0).values.tolist()))
mf_us = (rt[:, 3] - rt[:, 2]).median()
mf_cy = (cy[:, 2] - cy[:, 1]).median()
print(f"  mfma phase: {mf_us:.2f} us, {mf_cy:.0f} cycles -> {mf_cy / mf_us / 1e3:.2f} GHz; ideal 18432 cycles")
