"""BASELINE.json configs at the size bench.py times them.  The oracle cannot run these sizes in seconds, so the
full-size assertions are size-independent properties -- two independent HIP paths agreeing, determinism, the step counts -- and the
oracle is compared at the same BATCH on a grid it can afford.

configs[2] (dopri5 rtol 1e-5 + adjoint, B = 64, T = 10, t = arange(10, 20) / 20; atol 1e-5 = the reference's DiffEqSolver default,
modules/DiffEqSolver.py:13):
  * the device-controlled seminorm adjoint (csrc/adjoint_device.hip) against the host-driven loop (csrc/adjoint_dopri5.hip) on the
    bench's own inputs and on kink-free dynamics: same (nfe, accepted, rejected), gradients equal to round-off (the two sum the
    error-norm partials in different fixed orders, so a step size may differ in its last bit);
  * the oracle's adjoint at B = 64 on a 3-point grid, kink-free dynamics: gradients <= 1e-4.
configs[4] (bf16, B = 128, T = 40) lives in tests/test_hip_bf16.py (the (128, 40) cases)."""
import os

import pytest
import torch

from conftest import record, rel_l2

pytestmark = pytest.mark.gpu


def _bench_inputs(cuda, batch=64, T=10, kink_free=False):
    """bench.py's synthetic workload: default Conv2d init under manual_seed(0), z0 = randn(seed 1234) * 0.5.
    kink_free: the dynamics of tests/test_hip_backward.py instead (no pre-activation anywhere near a ReLU kink)."""
    import ode_rl_amd
    if kink_free:
        from test_hip_backward import _kink_free
        f = _kink_free()[0].to(cuda)
    else:
        torch.manual_seed(0)
        f = ode_rl_amd.ODEFunc(64, 64, 3, 64, False, "relu", final_act=False).to(cuda)
    g = torch.Generator().manual_seed(1234)
    z0 = (torch.randn(batch, 64, 16, 16, generator=g) * 0.5).to(cuda)
    t = torch.arange(T, 2 * T, dtype=torch.float64) / (2 * T)
    gout = torch.randn(T, batch, 64, 16, 16, generator=g).to(cuda)
    return f, z0, t, gout


def _adjoint_step(f, z0, t, gout, rtol, atol, norm):
    import ode_rl_amd
    f.zero_grad()
    z = z0.clone().requires_grad_(True)
    opts = {"norm": "seminorm"} if norm == "seminorm" else None
    sol = ode_rl_amd.odeint_adjoint(f, z, t, rtol=rtol, atol=atol, method="dopri5", adjoint_options=opts)
    fwd = dict(ode_rl_amd.last_stats)
    sol.backward(gout)
    st = dict(ode_rl_amd.last_adjoint_stats)
    return sol.detach().clone(), z.grad.clone(), [p.grad.clone() for p in f.parameters()], fwd, st


@pytest.mark.parametrize("kink_free", [False, True])
def test_config2_full_size_device_controller_equals_host_loop(cuda, kink_free):
    """Default-initialised ReLU dynamics (the bench's): the two paths sum the error-norm partials in different fixed orders, a step size
    may differ in its last bit, the states then differ by ~1e-7 and a pre-activation within that of 0 flips one ReLU-mask element --
    the weight gradients of the hidden layers agree to ~2e-5 (observed), inside the 1e-4 of every gradient comparison here.  On
    kink-free dynamics nothing can flip: <= 1e-6."""
    if os.environ.get("ODEHIP_PERSISTENT") == "0" or os.environ.get("ODEHIP_ADJOINT_DEVICE") == "0":
        pytest.skip("needs the device-controlled adjoint")
    f, z0, t, gout = _bench_inputs(cuda, kink_free=kink_free)
    tag = 'kinkfree' if kink_free else 'bench'
    dev = _adjoint_step(f, z0, t, gout, 1e-5, 1e-5, "seminorm")
    dev2 = _adjoint_step(f, z0, t, gout, 1e-5, 1e-5, "seminorm")
    os.environ["ODEHIP_ADJOINT_DEVICE"] = "0"
    try:
        host = _adjoint_step(f, z0, t, gout, 1e-5, 1e-5, "seminorm")
    finally:
        os.environ.pop("ODEHIP_ADJOINT_DEVICE")
    # the forward solve is the same code in both runs
    assert torch.equal(dev[0], host[0]) and dev[3]["nfe"] == host[3]["nfe"]
    # deterministic: bitwise
    assert torch.equal(dev[1], dev2[1]) and all(torch.equal(a, b) for a, b in zip(dev[2], dev2[2]))
    # one interval = one fresh solve: 2 evaluations for the initial step + 6 per attempted step
    for st in (dev[4], host[4]):
        assert st["nfe"] == 2 * (len(t) - 1) + 6 * (st["n_accept"] + st["n_reject"]) and st["n_accept"] >= len(t) - 1
    assert (dev[4]["nfe"], dev[4]["n_accept"], dev[4]["n_reject"]) == (host[4]["nfe"], host[4]["n_accept"], host[4]["n_reject"]), (dev[4], host[4])
    record(f"config2.B64.{tag}.n_accept", dev[4]["n_accept"])
    record(f"config2.B64.{tag}.n_reject", dev[4]["n_reject"])
    errs = [record(f"config2.B64.{tag}.dev_vs_host.grad_z0", rel_l2(dev[1], host[1]))]
    errs += [record(f"config2.B64.{tag}.dev_vs_host.grad_p{i}", rel_l2(a, b)) for i, (a, b) in enumerate(zip(dev[2], host[2]))]
    assert max(errs) <= (1e-6 if kink_free else 1e-4), errs
    assert all(bool(torch.isfinite(g).all()) for g in [dev[1]] + dev[2])


def test_config2_batch64_adjoint_against_the_oracle(cuda):
    """The oracle's adjoint at the config's batch on a grid it finishes in seconds (3 points = 2 backward solves), kink-free dynamics
    (tests/test_hip_backward.py): every gradient <= 1e-4, forward <= 1e-5.  At rtol 1e-5 the error estimate -- a cancellation of seven
    stages -- sits at fp32 round-off, so its ratio and with it dt_next depend on the summation order of the convolution (Winograd here,
    direct in the oracle): the accepted-step count may differ by <= 2 (as in test_config2_dopri5_adjoint_batch8_full_grid)."""
    import ode_rl_amd
    from oracle import reference_modules as rm
    from oracle import torchdiffeq_ref
    from test_hip_backward import _kink_free
    f, sd = _kink_free()
    g = torch.Generator().manual_seed(64)
    z0 = torch.randn(64, 64, 16, 16, generator=g) * 0.5
    t = torch.tensor([0.5, 0.55, 0.6], dtype=torch.float64)       # the config's spacing (1 / 20)
    gout = torch.randn(3, 64, 64, 16, 16, generator=g)
    ws, bs = rm.split_convnet_state(sd, "gradient_net.")
    ws = [w.clone().requires_grad_(True) for w in ws]
    bs = [b.clone().requires_grad_(True) for b in bs]
    stats = {}
    ref_sol, ref_gz, ref_gp = torchdiffeq_ref.odeint_adjoint(rm.ode_func(ws, bs), z0, t, ws + bs, gout, rtol=1e-5, atol=1e-5,
                                                             method="dopri5", stats=stats, adjoint_norm="seminorm")
    f = f.to(cuda)
    zd = z0.to(cuda).requires_grad_(True)
    sol = ode_rl_amd.odeint_adjoint(f, zd, t, rtol=1e-5, atol=1e-5, method="dopri5", adjoint_options={"norm": "seminorm"})
    assert record("config2.B64.oracle.forward", rel_l2(sol, ref_sol)) <= 1e-5
    sol.backward(gout.to(cuda))
    got = ode_rl_amd.last_adjoint_stats
    assert abs(got["n_accept"] - stats["n_accept"]) <= 2 and got["nfe"] == 2 * (len(t) - 1) + 6 * (got["n_accept"] + got["n_reject"])
    errs = [record("config2.B64.oracle.grad_z0", rel_l2(zd.grad, ref_gz))]
    convs = [m for m in f.gradient_net if isinstance(m, torch.nn.Conv2d)]
    for i, (c, gw, gb) in enumerate(zip(convs, ref_gp[:5], ref_gp[5:])):
        errs += [record(f"config2.B64.oracle.grad_w{i}", rel_l2(c.weight.grad, gw)), record(f"config2.B64.oracle.grad_b{i}", rel_l2(c.bias.grad, gb))]
    assert max(errs) <= 1e-4, errs


def test_config1_full_size_is_deterministic_and_batch_independent(cuda):
    """configs[1] at the bench's size (B = 64, T = 10, rk4, fp32; the direct oracle comparison is tests/test_hip_odeint.py): the
    trajectory of a sample does not depend on which other samples share the batch, and a second run is bitwise the same."""
    import ode_rl_amd
    f, z0, t, _ = _bench_inputs(cuda)
    with torch.no_grad():
        a = ode_rl_amd.odeint(f, z0, t, method="rk4")
        b = ode_rl_amd.odeint(f, z0, t, method="rk4")
        perm = torch.randperm(64, generator=torch.Generator().manual_seed(5)).to(cuda)
        c = ode_rl_amd.odeint(f, z0[perm].contiguous(), t, method="rk4")
    assert torch.equal(a, b) and torch.equal(a[:, perm], c) and bool(torch.isfinite(a).all()) and torch.equal(a[0], z0)


def test_config3_full_size_with_the_reference_default_solver(cuda):
    """configs[3] names the VidODE model, whose decoder solver defaults to dopri5 (configs.yaml:79; DiffEqSolver's rtol 1e-4 / atol 1e-5,
    modules/DiffEqSolver.py:13): VidODE latents (128 ch, f = 128 -> 64 -> 64 -> 128) at the per-GPU batch of 64, T = 10, directly against
    the oracle -- the same (nfe, accepted, rejected), increments <= 5e-5 (the dopri5 tolerance of tests/test_hip_odeint.py: torchdiffeq's
    dense-output coefficients cancel in fp32)."""
    import ode_rl_amd
    from oracle import reference_modules as rm
    from oracle import torchdiffeq_ref
    torch.manual_seed(0)
    f = ode_rl_amd.ODEFunc(128, 128, 2, 64, False, "relu", final_act=False)
    ws, bs = rm.split_convnet_state({k: v.detach().clone() for k, v in f.state_dict().items()}, "gradient_net.")
    z0 = torch.randn(64, 128, 16, 16, generator=torch.Generator().manual_seed(1234)) * 0.5
    t = torch.arange(10, 20, dtype=torch.float64) / 20
    stats = {}
    with torch.no_grad():
        sol = ode_rl_amd.odeint(f.to(cuda), z0.to(cuda), t, rtol=1e-4, atol=1e-5, method="dopri5").cpu()
        got = dict(ode_rl_amd.last_stats)
        ref = torchdiffeq_ref.odeint(rm.ode_func(ws, bs), z0, t, rtol=1e-4, atol=1e-5, method="dopri5", stats=stats)
    assert (got["nfe"], got["n_accept"], got["n_reject"]) == (stats["nfe"], stats["n_accept"], stats.get("n_reject", 0)), (got, stats)
    assert torch.equal(sol[0], z0)
    assert record("config3.dopri5.B64.T10.increment", rel_l2(sol[1:] - z0, ref[1:] - z0)) <= 5e-5
