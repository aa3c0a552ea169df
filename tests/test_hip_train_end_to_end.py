"""End-to-end training step of the ODEConvGRU wiring (reference models/ODEConvGRU.py:57-98 + train_test.py:204
`loss.backward()`): conv encoder -> ODEConvGRUCell -> DiffEqSolver -> conv decoder -> sigmoid -> MSE, every parameter's
gradient against torch.autograd through the oracle pipeline on the CPU (strided / transposed convs either side of the
path are torch modules in both).  Tolerance: rel-L2 <= 1e-3 per parameter gradient (MIOpen vs CPU convs of the harness
either side of the path, LeakyReLU kinks; the hot-path pieces alone are held to 1e-4 / 2e-4 in their own tests)."""
import argparse
import copy

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def _model(method):
    from ode_rl_amd.models.ODEConvGRU import ODEConvGRU
    torch.manual_seed(2)
    opt = argparse.Namespace(resolution=64, n_downs=2, conv_encoder_out_ch=64, in_channels=1, n_ode_layers=3,
                             neural_ode_n_units=64, neural_ode_decoder_out_ch=64, decode_diff_method=method, mem=False,
                             z_sample=False)
    m = ODEConvGRU(opt, torch.device("cpu"))
    alt = torch.where(torch.arange(64) % 2 == 0, 2.5, -2.5)
    with torch.no_grad():   # keep the ReLUs of both dynamics and of the head away from their kink
        for f in (m.ode_encoder_func, m.ode_decoder_func):
            for i in (0, 2, 4, 6):
                f.gradient_net[i].weight.mul_(0.15)
                f.gradient_net[i].bias.copy_(alt)
            f.gradient_net[8].weight.mul_(4.0)
        m.ode_convgru_cell.transform_z0[0].weight.mul_(0.3)
        m.ode_convgru_cell.transform_z0[0].bias.copy_(alt)
    return m


def _oracle_loss(m, frames, truth, t_obs, t_pred, method, rtol, atol):
    from oracle import reference_modules as rm
    from oracle import torchdiffeq_ref
    sd = dict(m.named_parameters())
    ws_e, bs_e = rm.split_convnet_state({k[len("ode_encoder_func."):]: v for k, v in sd.items() if k.startswith("ode_encoder_func.")},
                                        "gradient_net.")
    ws_d, bs_d = rm.split_convnet_state({k[len("ode_decoder_func."):]: v for k, v in sd.items() if k.startswith("ode_decoder_func.")},
                                        "gradient_net.")
    cell = {k[len("ode_convgru_cell.cgru_cell."):]: v for k, v in sd.items() if k.startswith("ode_convgru_cell.cgru_cell.")}
    head = {k[len("ode_convgru_cell.transform_z0."):]: v for k, v in sd.items() if k.startswith("ode_convgru_cell.transform_z0.")}
    b, t, c, h, w = frames.shape
    enc = m.conv_encoder(frames.view(b * t, c, h, w))
    enc = enc.view(b, t, *enc.shape[1:]).permute(1, 0, 2, 3, 4)
    mean, _, _ = rm.ode_convgru_encode(enc, t_obs, rm.ode_func(ws_e, bs_e), cell, head)
    sol = torchdiffeq_ref.odeint(rm.ode_func(ws_d, bs_d), mean, t_pred, rtol=rtol, atol=atol, method=method)
    t2, b2 = sol.shape[:2]
    pred = torch.sigmoid(m.conv_decoder(sol.reshape(t2 * b2, *sol.shape[2:])))
    pred = pred.view(t2, b2, *pred.shape[1:]).permute(1, 0, 2, 3, 4)
    return torch.nn.functional.mse_loss(pred, truth), pred


@pytest.mark.parametrize("method", ["rk4", "dopri5"])
def test_training_step_gradients_match_oracle_pipeline(cuda, method):
    ref = _model(method)
    dev = copy.deepcopy(ref).to(cuda)
    g = torch.Generator().manual_seed(4)
    frames = torch.rand(2, 3, 1, 64, 64, generator=g)
    truth = torch.rand(2, 3, 1, 64, 64, generator=g)
    ts = torch.arange(6, dtype=torch.float64) / 6
    loss_ref, pred_ref = _oracle_loss(ref, frames, truth, ts[:3], ts[3:], method, 1e-4, 1e-5)
    loss_ref.backward()

    pred = dev(frames.to(cuda), {"observed_tp": ts[:3].to(cuda), "tp_to_predict": ts[3:].to(cuda)})
    loss = dev.get_loss(pred, truth.to(cuda))
    assert rel_l2(pred, pred_ref.detach()) <= 1e-4
    assert abs(float(loss) - float(loss_ref)) <= 1e-5 * abs(float(loss_ref)) + 1e-7
    loss.backward()
    refp = dict(ref.named_parameters())
    bad = {}
    for name, p in dev.named_parameters():
        assert p.grad is not None, name
        e = rel_l2(p.grad, refp[name].grad)
        if e > 1e-3:
            bad[name] = e
    assert not bad, bad
    # one optimizer step changes the parameters in place; the next forward must see them (packed-weight caches refresh)
    opt = torch.optim.Adam(dev.parameters(), lr=1e-3)
    opt.step()
    with torch.no_grad():
        pred2 = dev(frames.to(cuda), {"observed_tp": ts[:3].to(cuda), "tp_to_predict": ts[3:].to(cuda)})
    assert float((pred2 - pred.detach()).abs().max()) > 0
