"""`FusedAdam`: torch.optim.Adam semantics (the reference's optimizer, train_test.py:24) with ONE HIP launch per step for all
parameter tensors (csrc/adam.hip) instead of torch's per-op foreach launches.  State layout and `state_dict()` are those of
torch.optim.Adam (`step`, `exp_avg`, `exp_avg_sq`), so the reference's pickled optimizer state (helpers/utils.py:218-222)
loads into it and vice versa."""
import ctypes

import torch

from . import _lib
from .hip_ops import _stream, require_device_tensor


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            steps = set()
            keep = []
            for p in ps:
                require_device_tensor(p, "parameter")
                if not p.is_contiguous() or p.grad.is_sparse:
                    raise RuntimeError("FusedAdam needs dense, contiguous parameters")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] = st["step"] + 1 if torch.is_tensor(st["step"]) else torch.tensor(float(st["step"]) + 1)
                steps.add(int(st["step"]))
                keep.append(p.grad.contiguous())
            if len(steps) != 1:
                raise RuntimeError("FusedAdam: parameters of one group must share their step count")
            n = len(ps)
            arr = lambda xs: (ctypes.c_void_p * n)(*[x.data_ptr() for x in xs])
            numel = (ctypes.c_longlong * n)(*[p.numel() for p in ps])
            _lib.check(lib.odehip_adam_step(arr(ps), arr(keep), arr([self.state[p]["exp_avg"] for p in ps]),
                                            arr([self.state[p]["exp_avg_sq"] for p in ps]), numel, n, float(group["lr"]),
                                            float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]),
                                            float(group["weight_decay"]), steps.pop(), _stream()))
            for p in ps:   # the kernel wrote through raw pointers: tell autograd (and the packed-weight caches keyed on _version)
                torch.autograd.graph.increment_version(p)
        return loss
