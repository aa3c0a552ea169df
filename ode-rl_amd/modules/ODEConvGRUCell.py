"""Drop-in for the reference's `modules/ODEConvGRUCell.py:9-78`: same constructor and attribute names (`ode_func`,
`cgru_cell`, `transform_z0`, `z0_dim`), whole reverse-time Euler + ConvGRU loop in ONE C-ABI call."""
import torch
import torch.nn as nn

import os

from .. import hip_ops
from ..odeint import conv_stack_of
from .ConvGRUCell import ConvGRUCell


DEBUG_NAN = os.environ.get("ODEHIP_DEBUG_NAN") is not None   # debug mode: reproduce the reference's NaN assertions


class ODEConvGRUCell(nn.Module):
    def __init__(self, ode_func, opt, resolution, ch, out_ch=None, device=None, kernel_size=5):
        super().__init__()
        self.ode_func = ode_func
        self.device = device
        self.z0_diffeq_solver = None
        if out_ch is None:
            out_ch = ch
        self.cgru_cell = ConvGRUCell(input_size=resolution, input_dim=ch, hidden_dim=ch, kernel_size=kernel_size,
                                     bias=True).to(device)
        self.z0_dim = out_ch
        self.transform_z0 = nn.Sequential(nn.Conv2d(ch, ch, 1, 1, 0), nn.ReLU(), nn.Conv2d(ch, out_ch * 2, 1, 1, 0)).to(device)

    def _packed(self):
        p = getattr(self, "_hip_enc", None)
        if p is None:
            p = hip_ops.PackedEncoder(conv_stack_of(self.ode_func), self.cgru_cell._packed(), self.transform_z0)
            object.__setattr__(self, "_hip_enc", p)
        return p

    def forward(self, inputs, timesteps, mask=None):
        """inputs (T,B,C,H,W) time-first -> (mean_z0, std_z0), each (B, z0_dim, H, W); std is |.| (reference :32-37)."""
        enc = self._packed()
        if torch.is_grad_enabled() and (inputs.requires_grad or any(p.requires_grad for p in hip_ops.encoder_params(enc))):
            from ..autograd import encode_with_grad
            return encode_with_grad(enc, inputs, timesteps)
        mean, std, _ = hip_ops.odeconvgru_encode(enc, inputs, timesteps)
        if DEBUG_NAN:   # the reference asserts on NaN inside its loop (:56,59); here once per call, and only on request (it syncs)
            assert not torch.isnan(mean).any() and not torch.isnan(std).any(), "NaN in the encoder output"
        return mean, std

    def run_ode_conv_gru(self, inputs, timesteps, run_backwards=True, mask=None):
        """Returns (last yi, latent_ys (B,T,C,H,W)) as the reference (:39-78): frames visited T-1 .. 0 (run_backwards, what
        forward() uses, :33) or 0 .. T-1; slot k of latent_ys is the state after the k-th visited frame.  Differentiable: under
        autograd the training kernels (csrc/convgru_backward.hip) take the gradient that arrives through latent_ys."""
        if torch.is_grad_enabled() and (inputs.requires_grad or any(p.requires_grad for p in self.parameters())):
            from ..autograd import encode_with_grad
            _, _, latent = encode_with_grad(self._packed(), inputs, timesteps, want_latent=True, run_backwards=run_backwards)
            return latent[:, -1], latent
        _, _, latent = hip_ops.odeconvgru_encode(self._packed(), inputs, timesteps, want_latent=True, run_backwards=run_backwards)
        return latent[:, -1], latent
