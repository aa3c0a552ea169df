"""Numerics experiment (CPU): the weight gradient of a 3x3 convolution in the Winograd F(2x2,3x3) domain,
    dg = G^T [ sum_tiles (B^T d B) .* (A dY A^T) ] G        (16 instead of 36 multiplies per 2x2 outputs),
against autograd (fp64 identity, fp32 error).  python tools/experiments/winograd_wgrad_check.py"""
import numpy as np
import torch
import torch.nn.functional as F

torch.manual_seed(0)
BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def wgrad_wino(x, gy, dt):
    bt, g, at = (torch.tensor(m, dtype=dt) for m in (BT, G, AT))
    xp = F.pad(x.to(dt), (1, 1, 1, 1))
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                      # (B, Ci, th, tw, 4, 4)
    V = torch.einsum("ij,bcxyjk,lk->bcxyil", bt, tiles, bt)         # B^T d B
    gt = gy.to(dt).unfold(2, 2, 2).unfold(3, 2, 2)                  # (B, Co, th, tw, 2, 2)
    W = torch.einsum("ji,boxyjk,kl->boxyil", at, gt, at)            # A dY A^T  (A = AT^T)
    dU = torch.einsum("boxyil,bcxyil->ocil", W, V)
    return torch.einsum("ia,ocij,jb->ocab", g, dU, g)               # G^T dU G


x = torch.randn(8, 64, 16, 16) * 0.5
w = (torch.randn(64, 64, 3, 3) / 24).double().requires_grad_(True)
gy = torch.randn(8, 64, 16, 16)
F.conv2d(x.double(), w, padding=1).backward(gy.double())
ref = w.grad
print("identity check (fp64):", float((wgrad_wino(x, gy, torch.float64) - ref).norm() / ref.norm()))
w32 = w.detach().float().requires_grad_(True)
F.conv2d(x, w32, padding=1).backward(gy)
print(f"direct fp32 autograd: rel-L2 vs fp64 = {float((w32.grad.double() - ref).norm() / ref.norm()):.3e}")
print(f"Winograd-domain fp32: rel-L2 vs fp64 = {float((wgrad_wino(x, gy, torch.float32).double() - ref).norm() / ref.norm()):.3e}")
