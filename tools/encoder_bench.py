"""ODEConvGRUCell.forward alone (BASELINE shape: B=64, T_in=10, 64 channels) -- timing and, under rocprofv3, the kernel breakdown.
  python tools/encoder_bench.py [--batch 64] [--frames 10] [--steps 20] [--dtype f32|bf16]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--batch", type=int, default=64)
    p.add_argument("--frames", type=int, default=10)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--channels", type=int, default=64)
    p.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    a = p.parse_args()
    import ode_rl_amd
    if a.dtype == "bf16":
        ode_rl_amd.set_compute_dtype("bf16")
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    C = a.channels
    f = ode_rl_amd.ODEFunc(C, C, 3 if C == 64 else 2, 64, False, "relu", final_act=False)
    enc = ode_rl_amd.ODEConvGRUCell(f, None, (16, 16), C).to(dev)
    x = torch.randn(a.frames, a.batch, C, 16, 16, device=dev) * 0.5
    t = torch.arange(a.frames, dtype=torch.float64, device=dev) / (2 * a.frames)
    with torch.no_grad():
        for _ in range(3):
            enc(x, t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            enc(x, t)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    print(json.dumps({"what": "ODEConvGRUCell.forward", "batch": a.batch, "frames": a.frames, "channels": C, "dtype": a.dtype, "ms": ms}))


if __name__ == "__main__":
    main()
