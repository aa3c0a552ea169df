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
wt = torch.randn(64, 64, 3, 3, device=dev) / 24
w = hip_ops.pack_conv_weight(wt)
ww = hip_ops.pack_conv_weight_winograd(wt) if hip_ops.USE_WINOGRAD else None
bias = torch.randn(64, device=dev)
dst = torch.empty_like(x)
lib = _lib.load()
d = _lib.ConvDesc(src1=x.data_ptr(), src2=None, cin1=64, cin=64, cout=64, ks=3, batch=B, w_packed=w.data_ptr(), w_wino=ww.data_ptr() if ww is not None else None,
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
lib.odehip_set_debug_flags(8 | (16 if os.environ.get("STAMP_PRODUCER") else 0))
lib.odehip_debug_repeat_conv(ctypes.byref(d), 20, stream)
torch.cuda.synchronize()
lib.odehip_set_debug_flags(0)
lib.odehip_set_debug_buffer(None)
s = dbg.view(-1, 8).cpu().double()
names = ["(start, 100MHz ticks)", "first DMAs issued", "stage 0 landed", "stage 0 MFMAs done", "stage 1 MFMAs done",
         "stage 2 MFMAs done", "stage 3 MFMAs done", "end (stores drained)"]
print("per-workgroup stamps of the last launch, shader cycles since the workgroup's start: median [min..max]")
for i in range(1, 8):
    c = s[:, i]
    print(f"  {names[i]:>22}: {c.median():8.0f} [{c.min():8.0f} .. {c.max():8.0f}]")
print(f"  workgroup start skew: {(s[:, 0].max() - s[:, 0].min()) * 0.01:.2f} us; ideal MFMA cycles per stage: {72 * 65}")
