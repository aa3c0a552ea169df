"""GPU parity of the backward pass of the fixed-grid solvers against autograd through the oracle (the reference's
backward IS autograd through the solver's ops: modules/DiffEqSolver.py:9, train_test.py:204).
Tolerance: rel-L2 <= 1e-4 per gradient tensor (fp32; observed ~1e-6)."""
import pytest
import torch

from conftest import record, rel_l2

pytestmark = pytest.mark.gpu


def _setup(seed=0):
    import ode_rl_amd
    torch.manual_seed(seed)
    f = ode_rl_amd.ODEFunc(64, 64, 3, 64, False, "relu", final_act=False)
    sd = {k: v.detach().clone() for k, v in f.state_dict().items()}
    return f, sd


def _oracle_grads(sd, z0, t, gout, method, **kw):
    """Autograd through the restatement.  Also returns the ReLU margin: the smallest |pre-activation| met in any
    evaluation of f.  ReLU's derivative jumps at 0, so an input whose margin is below fp32 round-off can legitimately
    flip one mask element between two correct implementations; the test picks inputs with a safe margin."""
    import torch.nn.functional as F
    from oracle import reference_modules as rm
    from oracle import torchdiffeq_ref
    ws, bs = rm.split_convnet_state(sd, "gradient_net.")
    ws = [w.clone().requires_grad_(True) for w in ws]
    bs = [b.clone().requires_grad_(True) for b in bs]
    margin = [float("inf")]

    def f(tt, y):
        x = y
        for i, (w, b) in enumerate(zip(ws, bs)):
            x = F.conv2d(x, w, b, padding=1)
            if i < len(ws) - 1:
                margin[0] = min(margin[0], float(x.detach().abs().min()))
                x = torch.relu(x)
        return x
    z = z0.clone().requires_grad_(True)
    sol = torchdiffeq_ref.odeint(f, z, t, method=method, **kw)
    grads = torch.autograd.grad(sol, [z] + ws + bs, gout)
    return sol.detach(), grads[0], grads[1:1 + len(ws)], grads[1 + len(ws):], margin[0]



def _case(seed, T, batch):
    g = torch.Generator().manual_seed(seed)
    z0 = torch.randn(batch, 64, 16, 16, generator=g) * 0.5
    t = torch.tensor([0.1, 0.25, 0.3, 0.7][:T], dtype=torch.float64)
    gout = torch.randn(T, batch, 64, 16, 16, generator=g)
    return z0, t, gout


def _check(cuda, f, z0, t, gout, method, ref, tol):
    import ode_rl_amd
    ref_sol, ref_gz, ref_gw, ref_gb, _ = ref
    f = f.to(cuda)
    f.zero_grad()
    zd = z0.to(cuda).requires_grad_(True)
    sol = ode_rl_amd.odeint(f, zd, t, method=method)
    assert sol.requires_grad
    assert record(f"bwd.{method}.forward.increment", rel_l2(sol.detach().cpu()[1:] - z0, ref_sol[1:] - z0)) <= 3e-6
    sol.backward(gout.to(cuda))
    assert rel_l2(zd.grad, ref_gz) <= tol
    convs = [m for m in f.gradient_net if isinstance(m, torch.nn.Conv2d)]
    for c, gw, gb in zip(convs, ref_gw, ref_gb):
        assert rel_l2(c.weight.grad, gw) <= tol
        assert rel_l2(c.bias.grad, gb) <= tol


@pytest.mark.parametrize("method,T", [("rk4", 4), ("euler", 3), ("midpoint", 3), ("rk4", 2), ("midpoint", 2)])
def test_backward_strict_on_kink_free_dynamics(cuda, method, T):
    """Hidden biases of +-2.5 on alternating channels keep every pre-activation far from the ReLU kink (half the
    channels always active, half always masked), so the gradient is a smooth function and two correct fp32
    implementations must agree to round-off: rel-L2 <= 1e-4 on every gradient tensor."""
    f, _ = _setup()
    with torch.no_grad():
        for i in (0, 2, 4, 6):
            f.gradient_net[i].weight.mul_(0.15)   # keeps the conv part of each pre-activation well inside +-2.5
            f.gradient_net[i].bias.copy_(torch.where(torch.arange(64) % 2 == 0, 2.5, -2.5))
        f.gradient_net[8].weight.mul_(4.0)
    sd = {k: v.detach().clone() for k, v in f.state_dict().items()}
    z0, t, gout = _case(7, T, 3)
    ref = _oracle_grads(sd, z0, t, gout, method)
    assert ref[4] > 0.5, ref[4]
    _check(cuda, f, z0, t, gout, method, ref, 1e-4)


def test_backward_random_dynamics(cuda):
    """Default-initialised f: first a small case whose ReLU margin is safe (strict 1e-4); then the usual size, where a
    pre-activation within fp32 round-off of 0 may flip one mask element between two correct implementations -- the
    gradients then differ by ~1/sqrt(#elements), so that leg uses 2e-2 unless its margin happens to be safe."""
    f, sd = _setup()
    for seed in range(7, 60):
        z0, t, gout = _case(seed, 2, 1)
        ref = _oracle_grads(sd, z0, t, gout, "rk4")
        if ref[4] >= 1e-6:
            break
    else:
        raise AssertionError("no input with a safe ReLU margin found")
    _check(cuda, f, z0, t, gout, "rk4", ref, 1e-4)
    z0, t, gout = _case(7, 4, 3)
    ref = _oracle_grads(sd, z0, t, gout, "rk4")
    _check(cuda, f, z0, t, gout, "rk4", ref, 1e-4 if ref[4] >= 1e-6 else 2e-2)


def test_backward_is_deterministic_and_accumulates(cuda):
    import ode_rl_amd
    f, _ = _setup(1)
    f = f.to(cuda)
    z0 = torch.randn(4, 64, 16, 16, device=cuda) * 0.5
    t = torch.tensor([0.0, 0.2, 0.5], dtype=torch.float64)
    gout = torch.randn(3, 4, 64, 16, 16, device=cuda)

    def run():
        f.zero_grad()
        z = z0.clone().requires_grad_(True)
        ode_rl_amd.odeint(f, z, t, method="rk4").backward(gout)
        return z.grad.clone(), [p.grad.clone() for p in f.parameters()]
    gz1, gp1 = run()
    gz2, gp2 = run()
    assert torch.equal(gz1, gz2) and all(torch.equal(a, b) for a, b in zip(gp1, gp2))   # no float atomics anywhere
    # a loss through DiffEqSolver: gradients flow to y0 and to every parameter
    solver = ode_rl_amd.DiffEqSolver(f, "rk4", device=cuda)
    f.zero_grad()
    z = z0.clone().requires_grad_(True)
    loss = solver(z, t.to(cuda)).pow(2).mean()
    loss.backward()
    assert z.grad is not None and all(p.grad is not None and torch.isfinite(p.grad).all() for p in f.parameters())


@pytest.mark.parametrize("method,T", [("rk4", 4), ("euler", 3), ("midpoint", 3), ("rk4", 2)])
def test_adjoint_matches_oracle_adjoint(cuda, method, T):
    """torchdiffeq odeint_adjoint semantics (one step of the same method per interval, backwards, on the augmented
    state; y reset to the stored trajectory) against the restatement, on kink-free dynamics: rel-L2 <= 1e-4."""
    import ode_rl_amd
    from oracle import reference_modules as rm
    from oracle import torchdiffeq_ref
    f, _ = _setup()
    with torch.no_grad():
        for i in (0, 2, 4, 6):
            f.gradient_net[i].weight.mul_(0.15)
            f.gradient_net[i].bias.copy_(torch.where(torch.arange(64) % 2 == 0, 2.5, -2.5))
        f.gradient_net[8].weight.mul_(4.0)
    sd = {k: v.detach().clone() for k, v in f.state_dict().items()}
    z0, t, gout = _case(7, T, 3)
    ws, bs = rm.split_convnet_state(sd, "gradient_net.")
    ws = [w.clone().requires_grad_(True) for w in ws]
    bs = [b.clone().requires_grad_(True) for b in bs]
    ref_sol, ref_gz, ref_gp = torchdiffeq_ref.odeint_adjoint(rm.ode_func(ws, bs), z0, t, ws + bs, gout, method=method)
    f = f.to(cuda)
    zd = z0.to(cuda).requires_grad_(True)
    sol = ode_rl_amd.odeint_adjoint(f, zd, t, method=method)
    assert rel_l2(sol, ref_sol) <= 1e-4
    sol.backward(gout.to(cuda))
    assert rel_l2(zd.grad, ref_gz) <= 1e-4
    convs = [m for m in f.gradient_net if isinstance(m, torch.nn.Conv2d)]
    for c, gw, gb in zip(convs, ref_gp[:5], ref_gp[5:]):
        assert rel_l2(c.weight.grad, gw) <= 1e-4
        assert rel_l2(c.bias.grad, gb) <= 1e-4


@pytest.mark.parametrize("rtol,atol,T,batch", [(1e-3, 1e-4, 4, 3), (1e-5, 1e-6, 3, 3), (1e-3, 1e-4, 2, 3), (1e-3, 1e-4, 3, 20)])
def test_dopri5_adjoint_matches_oracle_adjoint(cuda, rtol, atol, T, batch):
    """Adaptive adjoint (BASELINE.json configs[2]): odeint_adjoint(method="dopri5", adjoint_options={"norm": "seminorm"})
    against the restatement of torchdiffeq's adjoint with the same norm, on kink-free dynamics.  Same accepted/rejected
    step sequence (counts equal) and rel-L2 <= 1e-4 on every gradient.  batch 20: above the sixteen-workgroup walk's limit, i.e. the
    four-workgroup adaptive walk the B = 64 configurations run on (its elementwise rows, 4 x 64 partials per sample)."""
    import ode_rl_amd
    from oracle import reference_modules as rm
    from oracle import torchdiffeq_ref
    f, _ = _setup()
    with torch.no_grad():
        for i in (0, 2, 4, 6):
            f.gradient_net[i].weight.mul_(0.15)
            f.gradient_net[i].bias.copy_(torch.where(torch.arange(64) % 2 == 0, 2.5, -2.5))
        f.gradient_net[8].weight.mul_(4.0)
    sd = {k: v.detach().clone() for k, v in f.state_dict().items()}
    z0, t, gout = _case(7, T, batch)
    ws, bs = rm.split_convnet_state(sd, "gradient_net.")
    ws = [w.clone().requires_grad_(True) for w in ws]
    bs = [b.clone().requires_grad_(True) for b in bs]
    stats = {}
    ref_sol, ref_gz, ref_gp = torchdiffeq_ref.odeint_adjoint(rm.ode_func(ws, bs), z0, t, ws + bs, gout, rtol=rtol, atol=atol,
                                                             method="dopri5", stats=stats, adjoint_norm="seminorm")
    f = f.to(cuda)
    zd = z0.to(cuda).requires_grad_(True)
    sol = ode_rl_amd.odeint_adjoint(f, zd, t, rtol=rtol, atol=atol, method="dopri5", adjoint_options={"norm": "seminorm"})
    assert rel_l2(sol, ref_sol) <= 1e-4
    sol.backward(gout.to(cuda))
    got = ode_rl_amd.last_adjoint_stats
    if rtol >= 1e-3:
        assert (got["nfe"], got["n_accept"], got["n_reject"]) == (stats["nfe"], stats["n_accept"], stats.get("n_reject", 0)), (got, stats)
    else:
        # at rtol 1e-5 the error estimate (a cancellation of seven stages) sits at fp32 round-off: its ratio, hence dt_next,
        # depends on the conv kernel's summation order (Winograd vs direct), so only the result is compared, not the step count
        assert abs(got["n_accept"] - stats["n_accept"]) <= 2 and got["nfe"] == 2 * (len(t) - 1) + 6 * (got["n_accept"] + got["n_reject"])
    assert rel_l2(zd.grad, ref_gz) <= 1e-4
    convs = [m for m in f.gradient_net if isinstance(m, torch.nn.Conv2d)]
    for c, gw, gb in zip(convs, ref_gp[:5], ref_gp[5:]):
        assert rel_l2(c.weight.grad, gw) <= 1e-4
        assert rel_l2(c.bias.grad, gb) <= 1e-4


def test_dopri5_adjoint_close_to_true_gradient(cuda):
    """The adaptive adjoint at tight tolerances approaches the exact gradient of the continuous ODE; so does autograd
    through a tight rk4 solve of the restatement.  rel-L2 <= 2e-3 between the two."""
    import ode_rl_amd
    f, _ = _setup()
    with torch.no_grad():
        for i in (0, 2, 4, 6):
            f.gradient_net[i].weight.mul_(0.15)
            f.gradient_net[i].bias.copy_(torch.where(torch.arange(64) % 2 == 0, 2.5, -2.5))
        f.gradient_net[8].weight.mul_(4.0)
    sd = {k: v.detach().clone() for k, v in f.state_dict().items()}
    z0, t, gout = _case(11, 3, 2)
    fine = torch.linspace(0, 1, 9, dtype=torch.float64)
    tf = torch.cat([t[0] + (t[1] - t[0]) * fine, t[1] + (t[2] - t[1]) * fine[1:]])
    gfine = torch.zeros(len(tf), *gout.shape[1:])
    gfine[0], gfine[8], gfine[16] = gout[0], gout[1], gout[2]
    ref = _oracle_grads(sd, z0, tf, gfine, "rk4")
    f = f.to(cuda)
    zd = z0.to(cuda).requires_grad_(True)
    sol = ode_rl_amd.odeint_adjoint(f, zd, t, rtol=1e-6, atol=1e-7, method="dopri5", adjoint_options={"norm": "seminorm"})
    sol.backward(gout.to(cuda))
    assert rel_l2(zd.grad, ref[1]) <= 2e-3
    convs = [m for m in f.gradient_net if isinstance(m, torch.nn.Conv2d)]
    for c, gw, gb in zip(convs, ref[2], ref[3]):
        assert rel_l2(c.weight.grad, gw) <= 2e-3
        assert rel_l2(c.bias.grad, gb) <= 2e-3


def _kink_free():
    f, _ = _setup()
    with torch.no_grad():
        for i in (0, 2, 4, 6):
            f.gradient_net[i].weight.mul_(0.15)
            f.gradient_net[i].bias.copy_(torch.where(torch.arange(64) % 2 == 0, 2.5, -2.5))
        f.gradient_net[8].weight.mul_(4.0)
    return f, {k: v.detach().clone() for k, v in f.state_dict().items()}


@pytest.mark.parametrize("rtol,atol,T,first_step,batch", [(1e-3, 1e-4, 4, 0.02, 3), (1e-5, 1e-6, 3, 0.01, 3), (1e-4, 1e-5, 2, 0.3, 3), (1e-4, 1e-5, 2, 0.3, 20)])
def test_dopri5_backward_matches_autograd_through_oracle(cuda, rtol, atol, T, first_step, batch):
    """loss.backward() through odeint(method="dopri5") -- the reference's default training path -- against autograd
    through the restatement.  options={"first_step": dt} makes every step size a constant of the graph, so the two must
    agree to round-off: rel-L2 <= 1e-4 on every gradient (kink-free dynamics).  (0.3 forces rejected first attempts.)"""
    import ode_rl_amd
    f, sd = _kink_free()
    z0, t, gout = _case(7, T, batch)   # (batch 20: the four-workgroup adaptive walk of the B = 64 configurations; <= 16: the sixteen-workgroup one)
    stats = {}
    ref = _oracle_grads(sd, z0, t, gout, "dopri5", rtol=rtol, atol=atol, options={"first_step": first_step}, stats=stats)
    assert ref[4] > 0.5
    f = f.to(cuda)
    zd = z0.to(cuda).requires_grad_(True)
    sol = ode_rl_amd.odeint(f, zd, t, rtol=rtol, atol=atol, method="dopri5", options={"first_step": first_step})
    assert sol.requires_grad
    assert rel_l2(sol, ref[0]) <= 1e-4
    got = ode_rl_amd.last_stats
    assert (got["n_accept"], got["n_reject"]) == (stats["n_accept"], stats.get("n_reject", 0)), (got, stats)
    sol.backward(gout.to(cuda))
    assert rel_l2(zd.grad, ref[1]) <= 1e-4
    convs = [m for m in f.gradient_net if isinstance(m, torch.nn.Conv2d)]
    for c, gw, gb in zip(convs, ref[2], ref[3]):
        assert rel_l2(c.weight.grad, gw) <= 1e-4
        assert rel_l2(c.bias.grad, gb) <= 1e-4


def test_dopri5_backward_default_first_step(cuda):
    """Without first_step the first dt comes from _select_initial_step and, in torchdiffeq, carries a gradient of its own
    (a term of the size of the local error that the HIP path does not differentiate; observed deviation ~1e-6):
    rel-L2 <= 1e-4 through the DiffEqSolver module with its rtol 1e-4 / atol 1e-5 defaults."""
    import ode_rl_amd
    f, sd = _kink_free()
    z0, t, gout = _case(9, 4, 3)
    ref = _oracle_grads(sd, z0, t, gout, "dopri5", rtol=1e-4, atol=1e-5)
    f = f.to(cuda)
    solver = ode_rl_amd.DiffEqSolver(f, "dopri5", device=cuda)
    zd = z0.to(cuda).requires_grad_(True)
    sol = solver(zd, t.to(cuda))
    assert rel_l2(sol, ref[0]) <= 1e-4
    sol.backward(gout.to(cuda))
    errs = [rel_l2(zd.grad, ref[1])]
    convs = [m for m in f.gradient_net if isinstance(m, torch.nn.Conv2d)]
    for c, gw, gb in zip(convs, ref[2], ref[3]):
        errs += [rel_l2(c.weight.grad, gw), rel_l2(c.bias.grad, gb)]
    print("dopri5 default-first-step gradient deviations:", ["%.2e" % e for e in errs])
    assert max(errs) <= 1e-4, errs


@pytest.mark.parametrize("rtol,atol,T", [(1e-3, 1e-4, 4), (1e-4, 1e-5, 3)])
def test_dopri5_adjoint_mixed_norm_matches_oracle(cuda, rtol, atol, T):
    """torchdiffeq's DEFAULT adjoint norm (mixed: max over y, a_y and every parameter tensor's RMS): the parameter block
    steers the backward steps.  Same step sequence as the restatement and rel-L2 <= 1e-4 on every gradient (kink-free f)."""
    import ode_rl_amd
    from oracle import reference_modules as rm
    from oracle import torchdiffeq_ref
    f, sd = _kink_free()
    z0, t, gout = _case(7, T, 3)
    ws, bs = rm.split_convnet_state(sd, "gradient_net.")
    ws = [w.clone().requires_grad_(True) for w in ws]
    bs = [b.clone().requires_grad_(True) for b in bs]
    # the oracle orders the parameter block (w0..w4, b0..b4); the norm is a max over tensors, so the order is immaterial
    stats = {}
    ref_sol, ref_gz, ref_gp = torchdiffeq_ref.odeint_adjoint(rm.ode_func(ws, bs), z0, t, ws + bs, gout, rtol=rtol, atol=atol,
                                                             method="dopri5", stats=stats)
    f = f.to(cuda)
    zd = z0.to(cuda).requires_grad_(True)
    sol = ode_rl_amd.odeint_adjoint(f, zd, t, rtol=rtol, atol=atol, method="dopri5")
    sol.backward(gout.to(cuda))
    got = ode_rl_amd.last_adjoint_stats
    assert (got["nfe"], got["n_accept"], got["n_reject"]) == (stats["nfe"], stats["n_accept"], stats.get("n_reject", 0)), (got, stats)
    assert rel_l2(zd.grad, ref_gz) <= 1e-4
    convs = [m for m in f.gradient_net if isinstance(m, torch.nn.Conv2d)]
    for c, gw, gb in zip(convs, ref_gp[:5], ref_gp[5:]):
        assert rel_l2(c.weight.grad, gw) <= 1e-4
        assert rel_l2(c.bias.grad, gb) <= 1e-4


def test_config2_dopri5_adjoint_batch8_full_grid(cuda):
    """BASELINE configs[2] as stated, at a batch the oracle's adjoint finishes in seconds: dopri5 rtol 1e-5 / atol 1e-6 forward +
    adaptive adjoint (seminorm) over the config's own grid (T=10, t = arange(10,20)/20), B=8, against the restatement of
    torchdiffeq's adjoint.  Kink-free dynamics (see above) so that the gradient is smooth: rel-L2 <= 1e-4 everywhere."""
    import ode_rl_amd
    from oracle import reference_modules as rm
    from oracle import torchdiffeq_ref
    f, sd = _kink_free()
    g = torch.Generator().manual_seed(21)
    z0 = torch.randn(8, 64, 16, 16, generator=g) * 0.5
    t = torch.arange(10, 20, dtype=torch.float64) / 20
    gout = torch.randn(10, 8, 64, 16, 16, generator=g)
    ws, bs = rm.split_convnet_state(sd, "gradient_net.")
    ws = [w.clone().requires_grad_(True) for w in ws]
    bs = [b.clone().requires_grad_(True) for b in bs]
    stats = {}
    ref_sol, ref_gz, ref_gp = torchdiffeq_ref.odeint_adjoint(rm.ode_func(ws, bs), z0, t, ws + bs, gout, rtol=1e-5, atol=1e-6,
                                                             method="dopri5", stats=stats, adjoint_norm="seminorm")
    f = f.to(cuda)
    zd = z0.to(cuda).requires_grad_(True)
    sol = ode_rl_amd.odeint_adjoint(f, zd, t, rtol=1e-5, atol=1e-6, method="dopri5", adjoint_options={"norm": "seminorm"})
    assert record("config2.forward", rel_l2(sol, ref_sol)) <= 1e-5
    sol.backward(gout.to(cuda))
    got = ode_rl_amd.last_adjoint_stats
    assert abs(got["n_accept"] - stats["n_accept"]) <= 2 and got["nfe"] == 2 * (len(t) - 1) + 6 * (got["n_accept"] + got["n_reject"])
    errs = [record("config2.grad_z0", rel_l2(zd.grad, ref_gz))]
    convs = [m for m in f.gradient_net if isinstance(m, torch.nn.Conv2d)]
    for i, (c, gw, gb) in enumerate(zip(convs, ref_gp[:5], ref_gp[5:])):
        errs += [record(f"config2.grad_w{i}", rel_l2(c.weight.grad, gw)), record(f"config2.grad_b{i}", rel_l2(c.bias.grad, gb))]
    assert max(errs) <= 1e-4, errs


@pytest.mark.parametrize("first_step,batch", [(0.3, 3), (None, 64)])
def test_dopri5_saving_forward_equals_reintegration(cuda, first_step, batch):
    """Round 3: the forward of a dopri5 training step keeps the activations of the accepted steps in slots chosen ON THE DEVICE
    (rejected attempts reuse their slot -- first_step 0.3 forces rejections), the backward walks them without re-integrating.
    Against the re-integrating backward (ODEHIP_DOPRI5_SAVE=0) on the same inputs: same step sequence, gradients <= 1e-5 (the
    stage-2 input of a step is rounded differently by the two paths, nothing else differs); and a forward that accepts more steps
    than it has slots reports saved = False and still yields the re-integrating path's gradients bit for bit."""
    import os
    import ode_rl_amd
    from ode_rl_amd import hip_ops
    if os.environ.get("ODEHIP_PERSISTENT") == "0":
        pytest.skip("the saving forward needs the persistent walk")
    if first_step:   # the dynamics of test_hip_odeint.py::test_dopri5_forced_rejections: an oversized first step is rejected twice
        f, _ = _setup(3)
        with torch.no_grad():
            f.gradient_net[8].weight.mul_(12.0)
        first_step, rtol, atol = 3.0, 1e-4, 1e-5
    else:
        f, _ = _kink_free()
        rtol, atol = 1e-3, 1e-4
    f = f.to(cuda)
    g = torch.Generator().manual_seed(41)
    z0 = (torch.randn(batch, 64, 16, 16, generator=g) * 0.5).to(cuda)
    t = torch.tensor([0.0, 1.0, 2.5, 4.0], dtype=torch.float64)
    gout = torch.randn(4, batch, 64, 16, 16, generator=g).to(cuda)
    opts = {"first_step": first_step} if first_step else None

    def run():
        f.zero_grad()
        z = z0.clone().requires_grad_(True)
        sol = ode_rl_amd.odeint(f, z, t, rtol=rtol, atol=atol, method="dopri5", options=opts)
        st = dict(ode_rl_amd.last_stats)
        sol.backward(gout)
        return sol.detach().clone(), z.grad.clone(), [p.grad.clone() for p in f.parameters()], st

    old = os.environ.get("ODEHIP_DOPRI5_SAVE")
    slots = hip_ops._dopri5_save_slots
    try:
        os.environ["ODEHIP_DOPRI5_SAVE"] = "0"
        s0, gz0, gp0, st0 = run()
        os.environ.pop("ODEHIP_DOPRI5_SAVE")
        s1, gz1, gp1, st1 = run()
        assert st1["saved"] and not st0["saved"]
        assert (st1["nfe"], st1["n_accept"], st1["n_reject"]) == (st0["nfe"], st0["n_accept"], st0["n_reject"])
        if first_step:
            assert st1["n_reject"] >= 1 and st1["n_accept"] >= 2
        assert torch.equal(s1, s0)                       # the forward's arithmetic is untouched by saving
        assert record(f"dopri5.saved.grad_z0.B{batch}", rel_l2(gz1, gz0)) <= 1e-5
        for i, (a, b) in enumerate(zip(gp1, gp0)):
            assert record(f"dopri5.saved.grad_p{i}.B{batch}", rel_l2(a, b)) <= 1e-5
        s1b, gz1b, gp1b, _ = run()
        assert torch.equal(gz1b, gz1) and all(torch.equal(a, b) for a, b in zip(gp1b, gp1))   # deterministic
        if st1["n_accept"] >= 2:
            hip_ops._dopri5_save_slots = 1               # fewer slots than accepted steps
            s2, gz2, gp2, st2 = run()
            assert not st2["saved"] and torch.equal(s2, s0)
            assert torch.equal(gz2, gz0) and all(torch.equal(a, b) for a, b in zip(gp2, gp0))
            assert hip_ops._dopri5_save_slots >= st2["n_accept"]     # the next forward gets enough
    finally:
        hip_ops._dopri5_save_slots = slots
        if old is None:
            os.environ.pop("ODEHIP_DOPRI5_SAVE", None)
        else:
            os.environ["ODEHIP_DOPRI5_SAVE"] = old


def test_async_dopri5_forward_matches_the_synchronous_one(cuda):
    """With ode_rl_amd.set_async_dopri5(True) a dopri5 solve only enqueues its attempted steps; the outcome is read at the backward
    pass / at the first look into last_stats / at the next solve.  Same trajectory, same gradients (bit for bit), same stats as the
    synchronous call, and an error (max_num_steps) surfaces when the outcome is collected.

    Round 4: a solve that needs MORE attempts than were enqueued up front can no longer be completed behind its
    consumers' backs.  It is sealed on the device -- what a consumer enqueued between forward and collect reads is NaN in the frames
    the solve had not reached, never stale memory -- and collect() (here: backward) raises AsyncSolveTruncated, for odeint and for
    odeint_adjoint alike; the next solve enqueues more attempts and matches the synchronous one again."""
    import ode_rl_amd
    from ode_rl_amd import hip_ops
    from ode_rl_amd._lib import AsyncSolveTruncated
    f, _ = _setup(3)
    with torch.no_grad():
        f.gradient_net[8].weight.mul_(12.0)
    f = f.to(cuda)
    g = torch.Generator().manual_seed(41)
    z0 = (torch.randn(3, 64, 16, 16, generator=g) * 0.5).to(cuda)
    t = torch.tensor([0.0, 1.0, 2.5, 4.0], dtype=torch.float64)
    gout = torch.randn(4, 3, 64, 16, 16, generator=g).to(cuda)

    def forward(adjoint):
        f.zero_grad()
        z = z0.clone().requires_grad_(True)
        if adjoint:
            sol = ode_rl_amd.odeint_adjoint(f, z, t, rtol=1e-4, atol=1e-5, method="dopri5", options={"first_step": 3.0},
                                            adjoint_options={"norm": "seminorm"})
        else:
            sol = ode_rl_amd.odeint(f, z, t, rtol=1e-4, atol=1e-5, method="dopri5", options={"first_step": 3.0})
        return z, sol

    def run(adjoint=False):
        z, sol = forward(adjoint)
        sol.backward(gout)
        st = dict(ode_rl_amd.last_stats)
        return sol.detach().clone(), z.grad.clone(), [p.grad.clone() for p in f.parameters()], st

    ref = run()
    ref_adj = run(adjoint=True)
    assert ref[3]["n_reject"] >= 2 and ref[3]["n_accept"] >= 3
    with torch.no_grad():
        ref_inf = ode_rl_amd.odeint(f, z0, t, rtol=1e-4, atol=1e-5, method="dopri5")
        ref_inf_stats = dict(ode_rl_amd.last_stats)
    was = ode_rl_amd.set_async_dopri5(True)
    attempts = hip_ops._async_attempts
    try:
        for adjoint in (False, True):
            hip_ops._async_attempts = 2      # fewer than the solve needs (2 rejected + 3 accepted)
            z, sol = forward(adjoint)
            # a CONSUMER between the forward and the collect (the decoder and the loss of a training step): it must not see
            # uninitialised frames.  Frame 0 (= z0) is there, the last frame cannot have been reached by two attempts: NaN
            seen = sol.detach().clone()
            assert len(hip_ops._pending_solves) == 1
            with pytest.raises(AsyncSolveTruncated):
                sol.backward(gout)
            assert not hip_ops._pending_solves
            assert torch.equal(seen[0], z0) and bool(torch.isnan(seen[-1]).all())
            assert bool((torch.isnan(seen).flatten(1).all(1) | torch.isfinite(seen).flatten(1).all(1)).all())   # whole frames only
            assert hip_ops._async_attempts >= 4          # the next solve enqueues more ...
            with pytest.raises(AsyncSolveTruncated):     # ... 4 is still short of 5: sealed again, never silently completed
                run(adjoint)
            assert hip_ops._async_attempts >= 8
            got = run(adjoint)                           # ... and now the solve fits: bit-identical to the synchronous call
            want = ref_adj if adjoint else ref
            assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]) and all(torch.equal(a, b) for a, b in zip(got[2], want[2]))
            for k in ("nfe", "n_accept", "n_reject", "accepted"):
                assert got[3][k] == want[3][k], k
        with torch.no_grad():
            inf = ode_rl_amd.odeint(f, z0, t, rtol=1e-4, atol=1e-5, method="dopri5")
            assert len(hip_ops._pending_solves) == 1          # nothing has been waited for yet
            assert ode_rl_amd.last_stats["nfe"] == ref_inf_stats["nfe"] and not hip_ops._pending_solves
        assert torch.equal(inf, ref_inf)
        # a synchronous entry point first collects what is in flight (it shares the solver workspace with it)
        with torch.no_grad():
            inf = ode_rl_amd.odeint(f, z0, t, rtol=1e-4, atol=1e-5, method="dopri5")
            assert len(hip_ops._pending_solves) == 1
            from ode_rl_amd.odeint import conv_stack_of
            sync_out, _ = hip_ops.odeint_dopri5(conv_stack_of(f), z0, hip_ops.host_times(t), 1e-4, 1e-5)
            assert not hip_ops._pending_solves and torch.equal(sync_out, ref_inf) and torch.equal(inf, ref_inf)
        # an error is reported when the outcome is collected, as the exception the synchronous call raises
        with torch.no_grad():
            ode_rl_amd.odeint(f, z0, t, rtol=1e-4, atol=1e-5, method="dopri5", options={"first_step": 3.0, "max_num_steps": 2})
            with pytest.raises(AssertionError):
                ode_rl_amd.last_stats["nfe"]
            again = ode_rl_amd.odeint(f, z0, t, rtol=1e-4, atol=1e-5, method="dopri5")     # the library is usable afterwards
            ode_rl_amd.collect_pending_solves()
        assert torch.equal(again, ref_inf)
    finally:
        hip_ops._async_attempts = attempts
        ode_rl_amd.set_async_dopri5(was)
