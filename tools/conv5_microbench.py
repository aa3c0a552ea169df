"""The ConvGRU's two 5x5 convolutions (B=64: cat(x 64, h 64) -> 128 and -> 64 channels): direct kernel against Winograd F(2x2,5x5).
  python tools/conv5_microbench.py [batch]"""
import ctypes
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from ode_rl_amd import _lib, hip_ops  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 64
lib = _lib.load()
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for cout in (128, 64):
    x = hip_ops.nchw_to_q4(torch.randn(B, 64, 16, 16, device=dev))
    h = hip_ops.nchw_to_q4(torch.randn(B, 64, 16, 16, device=dev))
    wt = torch.randn(cout, 128, 5, 5, device=dev) / 56
    w, ww = hip_ops.pack_conv_weight(wt), hip_ops.pack_conv_weight_winograd5(wt)
    bias = torch.randn(cout, device=dev)
    dst = torch.empty(B, cout // 4, 256, 4, device=dev)
    flop = 2.0 * B * cout * 128 * 25 * 256
    for name, wino in (("direct", None), ("F(2x2,5x5)", ww)):
        d = _lib.ConvDesc(src1=x.data_ptr(), src2=h.data_ptr(), cin1=64, cin=128, cout=cout, ks=5, batch=B, w_packed=w.data_ptr(),
                          w_wino=wino.data_ptr() if wino is not None else None, bias=bias.data_ptr(), dst=dst.data_ptr(), relu=0)
        _lib.check(lib.odehip_debug_repeat_conv(ctypes.byref(d), 5, stream))
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(lib.odehip_debug_repeat_conv(ctypes.byref(d), 20, stream))
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        ts.sort()
        print(f"B={B} 128->{cout} {name:>11}: {ts[2]:7.1f} us per launch   {flop / ts[2] / 1e6:6.1f} algorithmic TFLOP/s")
        if wino is not None and "--ablate" in sys.argv:
            for nm, fl in (("no transform", 256), ("no DMA", 512), ("no transform, no DMA", 768), ("no fragment reads", 1024), ("no MFMA", 2048),
                           ("MFMA only", 256 | 512 | 1024)):
                lib.odehip_set_debug_flags(fl)
                _lib.check(lib.odehip_debug_repeat_conv(ctypes.byref(d), 5, stream))
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _lib.check(lib.odehip_debug_repeat_conv(ctypes.byref(d), 20, stream))
                e1.record()
                torch.cuda.synchronize()
                lib.odehip_set_debug_flags(0)
                print(f"      {nm:>22}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
