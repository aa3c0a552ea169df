"""Diagnostic: back-to-back launches of the ConvGRU 5x5 convs (cat(x,h) 128 -> 128 and 128 -> 64, B=64) under the ring kernel's
ablation flags (1: skip DMA, 2: skip MFMA, 4: skip epilogue).  Run on the GPU box: python tools/conv5_microbench.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ode_rl_amd  # noqa: E402,F401
from ode_rl_amd import hip_ops, _lib  # noqa: E402

dev = torch.device("cuda:0")
B = 64
lib = _lib.load()
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for cout in (128, 64):
    x = hip_ops.nchw_to_q4(torch.randn(B, 64, 16, 16, device=dev))
    h = hip_ops.nchw_to_q4(torch.randn(B, 64, 16, 16, device=dev))
    wt = torch.randn(cout, 128, 5, 5, device=dev) / 56
    w = hip_ops.pack_conv_weight(wt)
    bias = torch.randn(cout, device=dev)
    dst = torch.empty(B, cout // 4, 256, 4, device=dev)
    d = _lib.ConvDesc(src1=x.data_ptr(), src2=h.data_ptr(), cin1=64, cin=128, cout=cout, ks=5, batch=B, w_packed=w.data_ptr(),
                      w_wino=None, w_bf16=None, bias=bias.data_ptr(), dst=dst.data_ptr(), relu=0)
    flop = 2.0 * B * cout * 128 * 25 * 256
    for flags, name in ((0, "full"), (1, "no-dma"), (2, "no-mfma"), (4, "no-epilogue"), (3, "no-dma,no-mfma"), (0, "full")):
        lib.odehip_set_debug_flags(flags)
        lib.odehip_debug_repeat_conv(ctypes.byref(d), 20, stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.odehip_debug_repeat_conv(ctypes.byref(d), 100, stream)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 100
        print(f"5x5 128->{cout} flags={flags} ({name:>14}): {us:7.2f} us/launch  {flop / us / 1e6:6.1f} TFLOP/s-equivalent")
    lib.odehip_set_debug_flags(0)
    wb = hip_ops.pack_conv_weight_bf16_ks(wt)
    d.w_bf16 = wb.data_ptr()
    lib.odehip_debug_repeat_conv(ctypes.byref(d), 20, stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.odehip_debug_repeat_conv(ctypes.byref(d), 100, stream)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 100
    print(f"5x5 128->{cout} bf16 operands                : {us:7.2f} us/launch  {flop / us / 1e6:6.1f} TFLOP/s-equivalent")
