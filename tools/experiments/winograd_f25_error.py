"""Numerics experiment (CPU): Winograd F(2x2,5x5) (points 0, +-1, +-2, inf: 36 instead of 100 multiplies per 2x2 outputs) in fp32
for the ConvGRU's conv shapes (128 -> 128 / 128 -> 64 on 16x16 maps, /root/reference/modules/ConvGRUCell.py:40-50), against an
fp64 direct convolution.  python tools/experiments/winograd_f25_error.py"""
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


def wino(x, w, dt):
    bt, g, at = (torch.tensor(m, dtype=dt) for m in (BT, G, AT))
    B_, C, H, W = x.shape
    xp = F.pad(x.to(dt), (2, 2, 2, 2))
    tiles = xp.unfold(2, 6, 2).unfold(3, 6, 2)                      # (B, C, th, tw, 6, 6)
    V = torch.einsum("ij,bcxyjk,lk->bcxyil", bt, tiles, bt)
    U = torch.einsum("ij,ocjk,lk->ocil", g, w.to(dt), g)
    M = torch.einsum("ocil,bcxyil->boxyil", U, V)
    Y = torch.einsum("ij,boxyjk,lk->boxyil", at, M, at)
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B_, w.shape[0], H, W)


x = torch.randn(4, 128, 16, 16) * 0.5
w = (torch.rand(128, 128, 5, 5) * 2 - 1) / np.sqrt(128 * 25)   # nn.Conv2d's default init range
ref = F.conv2d(x.double(), w.double(), padding=2)
print("identity check (fp64):", float((wino(x, w, torch.float64) - ref).norm() / ref.norm()))
for name, y in (("direct fp32", F.conv2d(x, w, padding=2)), ("F(2x2,5x5) fp32", wino(x, w, torch.float32))):
    print(f"{name:>16}: rel-L2 vs fp64 = {float((y.double() - ref).norm() / ref.norm()):.3e}   max abs {float((y.double() - ref).abs().max()):.3e}")
