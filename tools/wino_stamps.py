import ctypes, os, sys, torch
sys.path.insert(0, '/root/repo')
import ode_rl_amd
from ode_rl_amd import hip_ops, _lib
dev = torch.device("cuda:0"); B = 64
x = hip_ops.nchw_to_q4(torch.randn(B, 64, 16, 16, device=dev))
wt = torch.randn(64, 64, 3, 3, device=dev) / 24
w = hip_ops.pack_conv_weight(wt); ww = hip_ops.pack_conv_weight_winograd(wt)
bias = torch.randn(64, device=dev); dst = torch.empty_like(x)
lib = _lib.load()
d = _lib.ConvDesc(src1=x.data_ptr(), src2=None, cin1=64, cin=64, cout=64, ks=3, batch=B, w_packed=w.data_ptr(), w_wino=ww.data_ptr(), bias=bias.data_ptr(), dst=dst.data_ptr(), relu=1)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
dbg = torch.zeros(B * 4 * 8, dtype=torch.int64, device=dev)
lib.odehip_set_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
for name, fl in (("consumer view, full", 8), ("consumer view, no producers", 8|1), ("producer view, full", 8|16), ("producer view, no mfma", 8|16|2), ("consumer view, producers DMA only", 8|32), ("consumer view, producers transform only", 8|128), ("  transform w/o LDS writes", 8|128|256), ("  transform w/o LDS reads", 8|128|512), ("  transform VALU only", 8|128|256|512)):
    lib.odehip_set_debug_flags(fl)
    lib.odehip_debug_repeat_conv(ctypes.byref(d), 10, stream); torch.cuda.synchronize()
    s = dbg.view(-1, 8).cpu().double()
    print(name, [int(s[:, i].median()) for i in range(1, 8)])
lib.odehip_set_debug_flags(0); lib.odehip_set_debug_buffer(None)
