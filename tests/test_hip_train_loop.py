"""the optimizer step and the training-step harness around the HIP path.
FusedAdam vs torch.optim.Adam (same arithmetic order: <= 1e-6 relative after 5 steps, state_dicts interchangeable);
train_batch (reference train_test.py:169-207, ODEConvGRU branch) lowers the loss; checkpoint pickle round trip in the
reference's format (helpers/utils.py:212-252)."""
import argparse
import copy

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def test_fused_adam_matches_torch_adam(cuda):
    from ode_rl_amd.optim import FusedAdam
    torch.manual_seed(0)
    shapes = [(64, 64, 3, 3), (64,), (128, 128, 5, 5), (1,), (7, 3)] * 8      # 40 tensors: more than one launch chunk
    a = [torch.nn.Parameter(torch.randn(s, device=cuda)) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    kw = dict(lr=8e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    oa, ob = FusedAdam(a, **kw), torch.optim.Adam(b, **kw)
    for it in range(5):
        for p, q in zip(a, b):
            g = torch.randn_like(p) * (1 + it)
            p.grad, q.grad = g.clone(), g.clone()
        oa.step()
        ob.step()
    for p, q in zip(a, b):
        assert rel_l2(p, q) <= 1e-6
    # the state is torch.optim.Adam's: load the fused state into a torch Adam and continue identically
    oc = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in a], **kw)
    oc.load_state_dict(copy.deepcopy(oa.state_dict()))   # load_state_dict may alias tensors that already have the right dtype/device
    for (p, q) in zip(a, oc.param_groups[0]["params"]):
        g = torch.randn_like(p)
        p.grad, q.grad = g.clone(), g.clone()
    oa.step()
    oc.step()
    for p, q in zip(a, oc.param_groups[0]["params"]):
        assert rel_l2(p, q) <= 1e-6


def test_train_batch_and_checkpoint_round_trip(cuda, tmp_path):
    from ode_rl_amd.models.ODEConvGRU import ODEConvGRU
    from ode_rl_amd.optim import FusedAdam
    from ode_rl_amd import train
    torch.manual_seed(1)
    opt = argparse.Namespace(resolution=64, n_downs=2, conv_encoder_out_ch=64, in_channels=1, n_ode_layers=3,
                             neural_ode_n_units=64, neural_ode_decoder_out_ch=64, decode_diff_method="dopri5", mem=False,
                             z_sample=False)
    model = ODEConvGRU(opt, torch.device("cpu")).to(cuda)
    optim = FusedAdam(model.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(2)
    ts = torch.arange(6, dtype=torch.float64) / 6
    batch = {"observed_data": torch.rand(2, 3, 1, 64, 64, generator=g) - 0.5, "data_to_predict": torch.rand(2, 3, 1, 64, 64, generator=g) - 0.5,
             "observed_tp": ts[:3].to(cuda), "tp_to_predict": ts[3:].to(cuda)}
    losses = []
    for _ in range(6):
        pred, truth, loss, ld = train.train_batch(model, batch, optim)
        losses.append(float(loss))
    assert pred.shape == (2, 3, 1, 64, 64) and float(pred.max()) <= 255.0 and truth.shape == pred.shape
    assert losses[-1] < losses[0], losses
    path = train.save_model_params(model, optim, epoch=0, step=6, logdir=str(tmp_path), ckpt_id="exp")
    assert path.endswith("exp_0000000006.pickle")
    fresh = ODEConvGRU(opt, torch.device("cpu")).to(cuda)
    opt2 = FusedAdam(fresh.parameters(), lr=1e-3)
    assert train.load_model_params(fresh, path, opt2) == (0, 6)
    with torch.no_grad():
        a = model(batch["observed_data"].to(cuda) + 0.5, batch)
        b = fresh(batch["observed_data"].to(cuda) + 0.5, batch)
    assert torch.equal(a, b)
    # both continue identically from the checkpoint
    _, _, la, _ = train.train_batch(model, batch, optim)
    _, _, lb, _ = train.train_batch(fresh, batch, opt2)
    assert abs(float(la) - float(lb)) <= 1e-6 * abs(float(la)) + 1e-9
