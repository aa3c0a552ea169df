"""Harness for end-to-end parity: the wiring of the reference's `models/ODEConvGRU.py:12-140`
(conv encoder -> ODEConvGRUCell -> DiffEqSolver -> conv decoder -> sigmoid) around the HIP hot path.

The strided encoder / transposed-conv decoder either side of the path are ONE fused HIP launch each
(csrc/frame_codec.hip: the 32x32 intermediate stays in LDS, the encoder writes time-first, the decoder reads the solver's
(T,B,C,16,16) as it lies and applies the sigmoid), with their own backward under autograd (csrc/frame_codec_backward.hip: one
frame channel, 32 / 64 latent channels); for other shapes under autograd, and for structures other than the reference's
n_downs = 2, they are library calls (MIOpen through torch).  Same module names so the reference's state_dict
loads, including the aliased keys (`diffeq_solver.ode_func.*` == `ode_decoder_func.*`, `ode_convgru_cell.ode_func.*` ==
`ode_encoder_func.*`)."""
import torch
import torch.nn as nn

from .. import hip_ops
from ..modules.DiffEqSolver import DiffEqSolver, ODEFunc
from ..modules.ODEConvGRUCell import ODEConvGRUCell


def _act(nonlinear):
    if nonlinear == 'relu':
        return nn.ReLU()
    if nonlinear == 'leaky_relu':
        return nn.LeakyReLU(negative_slope=0.2, inplace=True)
    raise NotImplementedError('Wrong activation function')


class Encoder(nn.Module):
    def __init__(self, n_inputs, out_ch, n_downs, nonlinear='relu'):
        super().__init__()
        act, chan = _act(nonlinear), 16
        layers = [nn.Conv2d(n_inputs, chan, 3, 2, 1), act]
        for _ in range(n_downs - 2):  # the reference overwrites `layers` here (:111-113); kept
            layers = [nn.Conv2d(chan, chan * 2, 3, 2, 1), act]
            chan *= 2
        layers += [nn.Conv2d(chan, out_ch, 3, 2, 1), act]
        self.encoder = nn.Sequential(*layers)

    def forward(self, x):
        return self.encoder(x)

    def _fused(self, x):
        """0: library calls; 1: the fused launch (no gradient wanted); 2: the fused launch under autograd (csrc/frame_codec_backward.hip)."""
        if not (x.is_cuda and tuple(x.shape[-2:]) == (64, 64) and hip_ops.frame_encoder_supported(self.encoder)):
            return 0
        if not (torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))):
            return 1
        ok = hip_ops.codec_backward_enabled() and not x.requires_grad and hip_ops.frame_encoder_backward_supported(self.encoder)
        return 2 if ok else 0

    def encode_time_first(self, frames):
        """frames (B,T,c,H,W) -> (T,B,C',H/4,W/4): `forward` on the flattened frames and the time-first view of
        ODEConvGRU.py:63-68, as one fused launch where that applies."""
        how = self._fused(frames)
        if how == 1:
            return hip_ops.frame_encode(self.encoder, frames)
        if how == 2:
            return hip_ops.frame_encode_autograd(self.encoder, frames)
        b, t, c, h, w = frames.size()
        enc = self.encoder(frames.view(b * t, c, h, w))
        _, c_, h_, w_ = enc.size()
        return enc.view(b, -1, c_, h_, w_).permute(1, 0, 2, 3, 4)


class Decoder(nn.Module):
    def __init__(self, n_inputs, out_ch, n_ups, nonlinear='relu'):
        super().__init__()
        if nonlinear != 'leaky_relu':  # the reference raises for 'relu' too (:126-128)
            raise NotImplementedError('Wrong activation function')
        act, chan = _act(nonlinear), 32
        layers = [nn.ConvTranspose2d(n_inputs, chan, 4, 2, 1), act]
        for _ in range(n_ups - 2):
            layers = [nn.ConvTranspose2d(chan, chan // 2, 4, 2, 1), act]
            chan //= 2
        layers += [nn.ConvTranspose2d(chan, out_ch, 4, 2, 1)]
        self.decoder = nn.Sequential(*layers)

    def forward(self, x):
        return self.decoder(x)

    def decode_sigmoid(self, sol_y):
        """sol_y (T,B,C,h,w) -> sigmoid(decoder(sol_y)) as (T,B,c,4h,4w) (ODEConvGRU.py:84-86), one fused launch where that applies."""
        wants_grad = torch.is_grad_enabled() and (sol_y.requires_grad or any(p.requires_grad for p in self.parameters()))
        if sol_y.is_cuda and tuple(sol_y.shape[-2:]) == (16, 16) and hip_ops.frame_decoder_supported(self.decoder):
            if not wants_grad:
                return hip_ops.frame_decode(self.decoder, sol_y, True)
            if hip_ops.codec_backward_enabled() and hip_ops.frame_decoder_backward_supported(self.decoder):
                return hip_ops.frame_decode_autograd(self.decoder, sol_y, True)
        t, b, c, h, w = sol_y.size()
        pred = torch.sigmoid(self.decoder(sol_y.view(b * t, c, h, w)))
        _, c3, h3, w3 = pred.size()
        return pred.view(t, b, c3, h3, w3)


class ODEConvGRU(nn.Module):
    def __init__(self, opt, device):
        super().__init__()
        self.opt, self.device = opt, device
        self.resize = 2 ** opt.n_downs
        res = (opt.resolution // self.resize, opt.resolution // self.resize)
        ch = opt.conv_encoder_out_ch
        self.conv_encoder = Encoder(opt.in_channels, ch, opt.n_downs, nonlinear='leaky_relu').to(device)
        self.ode_encoder_func = ODEFunc(n_inputs=ch, n_outputs=ch, n_layers=opt.n_ode_layers, n_units=opt.neural_ode_n_units,
                                        downsize=False, nonlinear='relu', device=device, final_act=False)
        self.ode_convgru_cell = ODEConvGRUCell(self.ode_encoder_func, opt, res, ch, device=device)
        self.ode_decoder_func = ODEFunc(n_inputs=ch, n_outputs=opt.neural_ode_decoder_out_ch, n_layers=opt.n_ode_layers,
                                        n_units=opt.neural_ode_n_units, downsize=False, nonlinear='relu', device=device,
                                        final_act=False)
        self.diffeq_solver = DiffEqSolver(self.ode_decoder_func, opt.decode_diff_method, device=device, memory=opt.mem)
        self.conv_decoder = Decoder(opt.neural_ode_decoder_out_ch, opt.in_channels, opt.n_downs, nonlinear='leaky_relu').to(device)

    def forward(self, inputs, batch_dict):
        b, t, c, h, w = inputs.size()
        observed_tp, tp_to_predict = batch_dict['observed_tp'], batch_dict['tp_to_predict']
        enc = self.conv_encoder.encode_time_first(inputs)  # time first
        first_point_mu, _ = self.ode_convgru_cell(enc, observed_tp)
        sol_y = self.diffeq_solver(first_point_mu, tp_to_predict)  # (T,B,C,H,W)
        return self.conv_decoder.decode_sigmoid(sol_y).permute(1, 0, 2, 3, 4)

    def get_prediction(self, inputs, batch_dict=None):
        return self(inputs, batch_dict)

    def get_loss(self, pred_frames, truth, loss='MSE'):
        b, t, c, h, w = truth.size()
        return nn.functional.mse_loss(pred_frames.reshape(b * t, c, h, w), truth.reshape(b * t, c, h, w))
