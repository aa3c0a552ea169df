"""Diagnostic: the measured-and-rejected F(4x4,3x3) prototype (tools/experiments/conv_f43.hip) back to
back against the production Winograd F(2x2,3x3) layer, B=64, 64 -> 64, with a parity check against torch's conv2d first.
NOT part of the product library or its public header: this script compiles the prototype into its own shared object
(tools/experiments/build/libodehip_f43.so, linked against the product library for its error plumbing) and binds it here.
  python tools/experiments/f43_microbench.py [B]"""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import ode_rl_amd  # noqa: E402,F401
from ode_rl_amd import hip_ops, _lib  # noqa: E402

SO = os.path.join(HERE, "build", "libodehip_f43.so")


def build():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    libdir = os.path.join(ROOT, "ode-rl_amd", "lib")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-Wno-unused-function",
                           "-I", os.path.join(ROOT, "ode-rl_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(HERE, "conv_f43.hip"), "-L", libdir, "-lodecgru_hip", f"-Wl,-rpath,{libdir}", "-o", SO])


if "--build-only" in sys.argv:
    build()
    sys.exit(0)
if not os.path.exists(SO):
    build()
_lib.load()
f43 = ctypes.CDLL(SO)
f43.odehip_f43_weight_floats.restype = ctypes.c_size_t
f43.odehip_f43_input_floats.restype = ctypes.c_size_t
f43.odehip_f43_input_floats.argtypes = [ctypes.c_int]
f43.odehip_pack_conv_weight_f43.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
f43.odehip_f43_transform_input.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
f43.odehip_conv_f43.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def f43_pack_weight(w):
    u = torch.empty(f43.odehip_f43_weight_floats(), dtype=torch.float32, device=w.device)
    _lib.check(f43.odehip_pack_conv_weight_f43(w.contiguous().data_ptr(), u.data_ptr(), 0, _st()))
    return u


def f43_transform_input(x_q4):
    b = x_q4.shape[0]
    v = torch.empty(f43.odehip_f43_input_floats(b), dtype=torch.float32, device=x_q4.device)
    _lib.check(f43.odehip_f43_transform_input(x_q4.data_ptr(), v.data_ptr(), b, _st()))
    return v


def f43_conv(v, u, bias, batch, relu=False, repeat=1):
    out = torch.empty((batch, 16, 256, 4), dtype=torch.float32, device=v.device)
    _lib.check(f43.odehip_conv_f43(v.data_ptr(), u.data_ptr(), bias.data_ptr(), out.data_ptr(), batch, int(relu), int(repeat), _st()))
    return out


dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = hip_ops.nchw_to_q4(torch.randn(B, 64, 16, 16, device=dev) * 0.5)
w = torch.randn(64, 64, 3, 3, device=dev) / 24
bias = torch.randn(64, device=dev)
u = f43_pack_weight(w)
v = f43_transform_input(x)
ref = torch.relu(torch.nn.functional.conv2d(hip_ops.q4_to_nchw(x).double(), w.double(), bias.double(), padding=1))
got = hip_ops.q4_to_nchw(f43_conv(v, u, bias, B, relu=True)).double()
err = float((got - ref).norm() / ref.norm())
assert err <= 1e-5, err
print(f"F(4x4,3x3) parity vs torch conv2d (fp64): rel-L2 {err:.2e}")
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


us = timed(lambda n: f43_conv(v, u, bias, B, relu=True, repeat=n))
print(f"F(4x4,3x3) matrix kernel : {us:7.2f} us/launch  {flop / us / 1e6:6.1f} TFLOP/s algorithmic")
vbuf = torch.empty_like(v)
us_t = timed(lambda n: [f43.odehip_f43_transform_input(x.data_ptr(), vbuf.data_ptr(), B, stream) for _ in range(n)])
print(f"input transform kernel   : {us_t:7.2f} us/launch")
wt = hip_ops.pack_conv_weight(w)
ww = hip_ops.pack_conv_weight_winograd(w)
dst = torch.empty_like(x)
d = _lib.ConvDesc(src1=x.data_ptr(), src2=None, cin1=64, cin=64, cout=64, ks=3, batch=B, w_packed=wt.data_ptr(), w_wino=ww.data_ptr(),
                  w_bf16=None, bias=bias.data_ptr(), dst=dst.data_ptr(), relu=1)
us2 = timed(lambda n: lib.odehip_debug_repeat_conv(ctypes.byref(d), n, stream))
print(f"F(2x2,3x3) production    : {us2:7.2f} us/launch  {flop / us2 / 1e6:6.1f} TFLOP/s algorithmic")
