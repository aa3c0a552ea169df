"""Diagnostic: the experimental F(4x4,3x3) matrix kernel (conv_f43.hip) back to back, against the production Winograd
F(2x2,3x3) layer, B=64, 64 -> 64.  python tools/f43_microbench.py [B]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ode_rl_amd  # noqa: E402,F401
from ode_rl_amd import hip_ops, _lib  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = hip_ops.nchw_to_q4(torch.randn(B, 64, 16, 16, device=dev) * 0.5)
w = torch.randn(64, 64, 3, 3, device=dev) / 24
bias = torch.randn(64, device=dev)
u = hip_ops.f43_pack_weight(w)
v = hip_ops.f43_transform_input(x)
lib = _lib.load()
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
flop = 2.0 * B * 64 * 64 * 9 * 256


def timed(fn):
    fn(20)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn(500)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 500


us = timed(lambda n: hip_ops.f43_conv(v, u, bias, B, relu=True, repeat=n))
print(f"F(4x4,3x3) matrix kernel : {us:7.2f} us/launch  {flop / us / 1e6:6.1f} TFLOP/s algorithmic")
vbuf = torch.empty_like(v)
us_t = timed(lambda n: [lib.odehip_f43_transform_input(x.data_ptr(), vbuf.data_ptr(), B, stream) for _ in range(n)])
print(f"input transform kernel   : {us_t:7.2f} us/launch")
wt = hip_ops.pack_conv_weight(w)
ww = hip_ops.pack_conv_weight_winograd(w)
dst = torch.empty_like(x)
d = _lib.ConvDesc(src1=x.data_ptr(), src2=None, cin1=64, cin=64, cout=64, ks=3, batch=B, w_packed=wt.data_ptr(), w_wino=ww.data_ptr(),
                  w_bf16=None, bias=bias.data_ptr(), dst=dst.data_ptr(), relu=1)
us2 = timed(lambda n: lib.odehip_debug_repeat_conv(ctypes.byref(d), n, stream))
print(f"F(2x2,3x3) production    : {us2:7.2f} us/launch  {flop / us2 / 1e6:6.1f} TFLOP/s algorithmic")
