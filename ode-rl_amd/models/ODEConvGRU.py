"""Harness for end-to-end parity: the wiring of the reference's `models/ODEConvGRU.py:12-140`
(conv encoder -> ODEConvGRUCell -> DiffEqSolver -> conv decoder -> sigmoid) around the HIP hot path.

The strided encoder / transposed-conv decoder either side of the path are library calls (MIOpen through torch), as
SURVEY.md section 8 (a11, f2) scopes them; same module names so the reference's state_dict loads, including the aliased
keys (`diffeq_solver.ode_func.*` == `ode_decoder_func.*`, `ode_convgru_cell.ode_func.*` == `ode_encoder_func.*`)."""
import torch
import torch.nn as nn

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
        enc = self.conv_encoder(inputs.view(b * t, c, h, w))
        _, c_, h_, w_ = enc.size()
        enc = enc.view(b, -1, c_, h_, w_).permute(1, 0, 2, 3, 4)  # time first
        first_point_mu, _ = self.ode_convgru_cell(enc, observed_tp)
        sol_y = self.diffeq_solver(first_point_mu, tp_to_predict)  # (T,B,C,H,W)
        t2, b2, c2, h2, w2 = sol_y.size()
        pred = torch.sigmoid(self.conv_decoder(sol_y.view(b2 * t2, c2, h2, w2)))
        _, c3, h3, w3 = pred.size()
        return pred.view(t2, b2, c3, h3, w3).permute(1, 0, 2, 3, 4)

    def get_prediction(self, inputs, batch_dict=None):
        return self(inputs, batch_dict)

    def get_loss(self, pred_frames, truth, loss='MSE'):
        b, t, c, h, w = truth.size()
        return nn.functional.mse_loss(pred_frames.reshape(b * t, c, h, w), truth.reshape(b * t, c, h, w))
