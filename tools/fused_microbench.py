"""Diagnostic: the whole-f bf16 launch (fstack_bf16.hip) under ablation flags.  python tools/fused_microbench.py [B]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ode_rl_amd  # noqa: E402
from ode_rl_amd import hip_ops, _lib  # noqa: E402
from ode_rl_amd.odeint import conv_stack_of  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ode_rl_amd.set_compute_dtype("bf16")
f = ode_rl_amd.ODEFunc(64, 64, 3, 64, False, "relu", final_act=False).to(dev)
stack = conv_stack_of(f)
y = torch.randn(B, 64, 16, 16, device=dev) * 0.5
lib = _lib.load()
import ctypes  # noqa: E402
x = hip_ops.nchw_to_q4(y)
out = torch.empty_like(x)
scratch = torch.empty(2 * x.numel(), device=dev)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for mode in ("bf16 fused", "bf16 per layer", "f32"):
    if mode == "bf16 per layer":
        hip_ops.USE_FUSED = False
        stack._cache.clear()
    if mode == "f32":
        ode_rl_amd.set_compute_dtype("f32")
    desc = stack.refresh()
    for flags, name in (((0, "full"), (1, "no weight DMA"), (2, "no MFMA, no reads"), (64, "MFMA on constant operands"), (3, "no DMA, no MFMA"), (65, "const MFMA, no DMA"), (0, "full"))
                        if mode == "bf16 fused" else ((0, "full"),)):
        lib.odehip_set_debug_flags(flags)
        _lib.check(lib.odehip_debug_repeat_f(ctypes.byref(desc), x.data_ptr(), out.data_ptr(), scratch.data_ptr(), B, 20, stream))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.odehip_debug_repeat_f(ctypes.byref(desc), x.data_ptr(), out.data_ptr(), scratch.data_ptr(), B, 200, stream))
        e1.record()
        torch.cuda.synchronize()
        print(f"{mode:>15} flags={flags:3d} ({name:>26}): {e0.elapsed_time(e1) * 1e3 / 200:7.2f} us per f evaluation (B={B})")
    lib.odehip_set_debug_flags(0)
