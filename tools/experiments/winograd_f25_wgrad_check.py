"""Numerics experiment (CPU): the weight gradient of a 5x5 convolution in the Winograd F(2x2,5x5) domain,
   dg[co][ci] = G^T [ sum over samples and tiles  (A dY A^T)[co] .* (B^T d B)[ci] ] G        (36 instead of 100 multiplies per 2x2 outputs),
against autograd in fp64 (identity) and in fp32 (error next to the direct fp32 sum).  python tools/experiments/winograd_f25_wgrad_check.py"""
import numpy as np
import torch
import torch.nn.functional as F

torch.manual_seed(0)
PTS = [0.0, 1.0, -1.0, 2.0, -2.0]
BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
               [0, 4, 0, -5, 0, 1]], dtype=np.float64)
G = np.zeros((6, 5))
for i, p in enumerate(PTS):
    n = np.prod([p - q for q in PTS if q != p])
    G[i] = [p ** k / n for k in range(5)]
G[5, 4] = 1.0
AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 1]], dtype=np.float64)


def wgrad_wino(x, gy, dt):
    bt, g, at = (torch.tensor(m, dtype=dt) for m in (BT, G, AT))
    xp = F.pad(x.to(dt), (2, 2, 2, 2))
    tiles = xp.unfold(2, 6, 2).unfold(3, 6, 2)                                # (B, Ci, th, tw, 6, 6)
    V = torch.einsum("ij,bcxyjk,lk->bcxyil", bt, tiles, bt)                 # B^T d B
    gt = gy.to(dt).unfold(2, 2, 2).unfold(3, 2, 2)                           # (B, Co, th, tw, 2, 2)
    W = torch.einsum("ji,boxyjk,kl->boxyil", at, gt, at)                     # A dY A^T  (A = AT^T)
    M = torch.einsum("boxyil,bcxyil->ocil", W, V)
    return torch.einsum("ji,ocjk,kl->ocil", g, M, g)                         # G^T M G


B_, Ci, Co = 4, 128, 64
x = torch.randn(B_, Ci, 16, 16) * 0.5
gy = torch.randn(B_, Co, 16, 16)
w = torch.zeros(Co, Ci, 5, 5, dtype=torch.float64, requires_grad=True)
F.conv2d(x.double(), w, padding=2).backward(gy.double())
ref = w.grad
print("identity check (fp64):", float((wgrad_wino(x, gy, torch.float64) - ref).norm() / ref.norm()))
w32 = torch.zeros(Co, Ci, 5, 5, requires_grad=True)
F.conv2d(x, w32, padding=2).backward(gy)
print("direct fp32      : rel-L2 vs fp64 = %.3e" % float((w32.grad.double() - ref).norm() / ref.norm()))
print("F(2x2,5x5) fp32  : rel-L2 vs fp64 = %.3e" % float((wgrad_wino(x, gy, torch.float32).double() - ref).norm() / ref.norm()))
