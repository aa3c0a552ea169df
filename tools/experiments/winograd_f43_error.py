"""Numerics experiment (CPU): error of Winograd F(2x2,3x3) vs F(4x4,3x3) in fp32 for the path's layer shape (64 -> 64, 16x16),
against an fp64 direct convolution.  Decides whether the 1.78x fewer multiplies of F(4x4,3x3) are affordable within the
1e-4 trajectory tolerance.  python tools/experiments/winograd_f43_error.py"""
import numpy as np
import torch
import torch.nn.functional as F

torch.manual_seed(0)


def mats(m):
    if m == 2:
        BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
        G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
        AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)
    else:  # F(4x4,3x3), Lavin & Gray
        BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                       [0, 4, 0, -5, 0, 1]], dtype=np.float64)
        G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                      [0, 0, 1]], dtype=np.float64)
        AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)
    return [torch.tensor(x, dtype=torch.float32) for x in (BT, G, AT)]


def wino(x, w, m):
    BT, G, AT = mats(m)
    a = m + 2
    B_, C, H, W = x.shape
    xp = F.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(2, a, m).unfold(3, a, m)                     # (B, C, th, tw, a, a)
    V = torch.einsum("ij,bcxyjk,lk->bcxyil", BT, tiles, BT)        # B^T d B
    U = torch.einsum("ij,ocjk,lk->ocil", G, w, G)                  # G g G^T
    M = torch.einsum("ocil,bcxyil->boxyil", U, V)                  # sum over ci, per transform position
    Y = torch.einsum("ij,boxyjk,lk->boxyil", AT, M, AT)            # A^T M A : (B, O, th, tw, m, m)
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B_, w.shape[0], H, W)


x = torch.randn(4, 64, 16, 16) * 0.5
w = torch.randn(64, 64, 3, 3) / 24
ref = F.conv2d(x.double(), w.double(), padding=1)
for name, y in (("direct fp32", F.conv2d(x, w, padding=1)), ("F(2x2,3x3) fp32", wino(x, w, 2)), ("F(4x4,3x3) fp32", wino(x, w, 4))):
    print(f"{name:>16}: rel-L2 vs fp64 = {float((y.double() - ref).norm() / ref.norm()):.3e}")
# error growth through a stack of 5 layers x 36 evaluations is roughly sqrt(180) x the per-layer figure
