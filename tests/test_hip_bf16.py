"""bf16-compute path (BASELINE.json configs[4]: bf16 operands on the matrix cores, fp32 accumulation, fp32 solver state)
against an emulation of exactly that arithmetic in the oracle (operands rounded to bf16, fp32 everything else):
one conv layer rel-L2 <= 1e-5 (bf16 products are exact in fp32); a stack / trajectory <= 1e-3 and gradients <= 5e-3, because
an activation within fp32 round-off of a bf16 rounding boundary may round the other way in two correct implementations;
and against the exact fp32 oracle to show the size of the bf16 error itself (<= 2e-2)."""
import os

import pytest
import torch

from conftest import record, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture
def bf16_mode():
    import ode_rl_amd
    ode_rl_amd.set_compute_dtype("bf16")
    yield
    ode_rl_amd.set_compute_dtype(None)


def _f(ch_in=64, units=64, seed=0):
    import ode_rl_amd
    torch.manual_seed(seed)
    f = ode_rl_amd.ODEFunc(ch_in, ch_in, 3 if ch_in == 64 else 2, units, False, "relu", final_act=False)
    return f, {k: v.detach().clone() for k, v in f.state_dict().items()}


@pytest.mark.parametrize("ch,units,batch", [(64, 64, 3), (128, 64, 2)])
def test_f_bf16_matches_emulation(cuda, bf16_mode, ch, units, batch):
    from ode_rl_amd import hip_ops
    from ode_rl_amd.odeint import conv_stack_of
    from oracle import reference_modules as rm
    f, sd = _f(ch, units)
    ws, bs = rm.split_convnet_state(sd, "gradient_net.")
    y = torch.randn(batch, ch, 16, 16, generator=torch.Generator().manual_seed(1)) * 0.5
    emu = rm.convnet_forward(y, ws, bs, compute_dtype="bf16")
    exact = rm.convnet_forward(y, ws, bs)
    f = f.to(cuda)
    # one layer: products of bf16 values are exact in fp32, only the summation order differs
    c0 = f.gradient_net[0]
    one = hip_ops.q4_to_nchw(hip_ops.conv_q4(hip_ops.nchw_to_q4(y.to(cuda)), hip_ops.pack_conv_weight(c0.weight), c0.bias.detach(),
                                             c0.out_channels, 3, relu=True, w_bf16=hip_ops.pack_conv_weight_bf16(c0.weight)))
    assert rel_l2(one, torch.relu(rm.convnet_forward(y, ws[:1], bs[:1], compute_dtype="bf16"))) <= 1e-5
    # the stack: a hidden activation within fp32 round-off of a bf16 rounding boundary may round the other way in two correct
    # implementations (a 2^-8 relative jump of that one input), so layers compound to ~1e-4
    out = hip_ops.convstack_forward(conv_stack_of(f), y.to(cuda))
    assert rel_l2(out, emu) <= 1e-3
    e = rel_l2(out, exact)
    assert 1e-4 <= e <= 2e-2, e          # really bf16 (not the fp32 kernels), and of the expected size


@pytest.mark.parametrize("method", ["rk4", "dopri5"])
def test_trajectory_and_gradients_bf16(cuda, bf16_mode, method):
    import ode_rl_amd
    from oracle import reference_modules as rm
    from oracle import torchdiffeq_ref
    f, _ = _f()
    with torch.no_grad():   # kink-free dynamics (see test_hip_backward.py)
        for i in (0, 2, 4, 6):
            f.gradient_net[i].weight.mul_(0.15)
            f.gradient_net[i].bias.copy_(torch.where(torch.arange(64) % 2 == 0, 2.5, -2.5))
        f.gradient_net[8].weight.mul_(4.0)
    sd = {k: v.detach().clone() for k, v in f.state_dict().items()}
    ws, bs = rm.split_convnet_state(sd, "gradient_net.")
    ws = [w.requires_grad_(True) for w in ws]
    bs = [b.requires_grad_(True) for b in bs]
    g = torch.Generator().manual_seed(5)
    z0 = (torch.randn(3, 64, 16, 16, generator=g) * 0.5).requires_grad_(True)
    t = torch.tensor([0.1, 0.25, 0.3, 0.7], dtype=torch.float64)
    gout = torch.randn(4, 3, 64, 16, 16, generator=g)
    kw = dict(rtol=1e-3, atol=1e-4, options={"first_step": 0.05}) if method == "dopri5" else {}
    ref = torchdiffeq_ref.odeint(rm.ode_func(ws, bs, compute_dtype="bf16"), z0, t, method=method, **kw)
    rg = torch.autograd.grad(ref, [z0] + ws + bs, gout)
    f = f.to(cuda)
    zd = z0.detach().to(cuda).requires_grad_(True)
    sol = ode_rl_amd.odeint(f, zd, t, method=method, **kw)
    assert rel_l2(sol, ref.detach()) <= 1e-3
    sol.backward(gout.to(cuda))
    assert rel_l2(zd.grad, rg[0]) <= 5e-3
    convs = [m for m in f.gradient_net if isinstance(m, torch.nn.Conv2d)]
    for c, gw, gb in zip(convs, rg[1:6], rg[6:]):
        assert rel_l2(c.weight.grad, gw) <= 5e-3 and rel_l2(c.bias.grad, gb) <= 5e-3


def test_autocast_selects_bf16_and_backward_keeps_it(cuda):
    """torch.autocast(dtype=bfloat16) is how a user of the reference asks for bf16; the backward pass, which runs outside
    the autocast region on the autograd thread, must use the same compute dtype as its forward."""
    import ode_rl_amd
    f, _ = _f()
    f = f.to(cuda)
    z0 = torch.randn(2, 64, 16, 16, device=cuda) * 0.5
    t = torch.tensor([0.0, 0.2, 0.5], dtype=torch.float64)
    gout = torch.randn(3, 2, 64, 16, 16, device=cuda)

    def run(ctx):
        f.zero_grad()
        z = z0.clone().requires_grad_(True)
        with ctx:
            sol = ode_rl_amd.odeint(f, z, t, method="rk4")
        sol.backward(gout)
        return sol.detach(), z.grad.clone(), [p.grad.clone() for p in f.parameters()]
    import contextlib
    a = run(torch.autocast(device_type="cuda", dtype=torch.bfloat16))
    ode_rl_amd.set_compute_dtype("bf16")
    try:
        b = run(contextlib.nullcontext())
    finally:
        ode_rl_amd.set_compute_dtype(None)
    c = run(contextlib.nullcontext())
    assert a[0].dtype == torch.float32
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and all(torch.equal(u, v) for u, v in zip(a[2], b[2]))
    assert not torch.equal(a[0], c[0]) and rel_l2(a[0], c[0]) <= 2e-2


def test_convgru_cell_and_encoder_bf16(cuda, bf16_mode):
    """The 5x5 ConvGRU convs on the bf16 ring kernel (forward and input gradients) and the bf16 3x3 encoder dynamics, against
    the same arithmetic emulated in the oracle: one cell step <= 1e-3 (two chained bf16 convs + GroupNorm), encoder outputs
    <= 2e-3, gradients <= 1e-2; and the bf16 error against the exact fp32 oracle stays <= 3e-2."""
    import ode_rl_amd
    from oracle import reference_modules as rm
    torch.manual_seed(3)
    f = ode_rl_amd.ODEFunc(64, 64, 3, 64, False, "relu", final_act=False)
    enc = ode_rl_amd.ODEConvGRUCell(f, None, (16, 16), 64)
    alt = torch.where(torch.arange(64) % 2 == 0, 2.5, -2.5)
    with torch.no_grad():
        for i in (0, 2, 4, 6):
            f.gradient_net[i].weight.mul_(0.15)
            f.gradient_net[i].bias.copy_(alt)
        f.gradient_net[8].weight.mul_(4.0)
        enc.transform_z0[0].weight.mul_(0.3)
        enc.transform_z0[0].bias.copy_(alt)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in enc.state_dict().items()}
    ws, bs = rm.split_convnet_state(sd, "ode_func.gradient_net.")
    cell = {k[len("cgru_cell."):]: v for k, v in sd.items() if k.startswith("cgru_cell.")}
    head = {k[len("transform_z0."):]: v for k, v in sd.items() if k.startswith("transform_z0.")}
    g = torch.Generator().manual_seed(11)
    x, h = torch.randn(2, 64, 16, 16, generator=g) * 0.5, torch.randn(2, 64, 16, 16, generator=g) * 0.5
    enc = enc.to(cuda)
    with torch.no_grad():
        _, out = enc.cgru_cell(input_tensor=x.to(cuda)[None], h_cur=h.to(cuda), seq_len=1)
        emu = rm.convgru_cell(x, h, cell, compute_dtype="bf16")
        exact = rm.convgru_cell(x, h, cell)
    assert rel_l2(out, emu) <= 1e-3
    e = rel_l2(out, exact)
    assert 1e-4 <= e <= 3e-2, e
    inputs = (torch.randn(3, 2, 64, 16, 16, generator=g) * 0.5).requires_grad_(True)
    t = torch.arange(3, dtype=torch.float64) / 8
    gm, gs = torch.randn(2, 64, 16, 16, generator=g), torch.randn(2, 64, 16, 16, generator=g)
    mean, std, _ = rm.ode_convgru_encode(inputs, t, rm.ode_func(ws, bs, compute_dtype="bf16"), cell, head, compute_dtype="bf16")
    names = list(sd)
    rg = torch.autograd.grad([mean, std], [inputs] + [sd[k] for k in names], [gm, gs])
    xd = inputs.detach().to(cuda).requires_grad_(True)
    m2, s2 = enc(xd, t.to(cuda))
    assert rel_l2(m2, mean.detach()) <= 2e-3 and rel_l2(s2, std.detach()) <= 2e-3
    torch.autograd.backward([m2, s2], [gm.to(cuda), gs.to(cuda)])
    assert rel_l2(xd.grad, rg[0]) <= 1e-2
    ref = dict(zip(names, rg[1:]))
    bad = {n: rel_l2(p.grad, ref[n]) for n, p in enc.named_parameters() if rel_l2(p.grad, ref[n]) > 1e-2}
    assert not bad, bad


def test_config4_bf16_39_intervals(cuda, bf16_mode):
    """BASELINE configs[4] as stated: bf16 compute, 20 -> 40 frames, i.e. T = 40 output points = 39 rk4 intervals (156 evaluations
    of f whose bf16 rounding accumulates in the fp32 state), trajectory AND gradients, B=2, against the oracle emulating the
    same arithmetic (bf16 operands, fp32 accumulate / state).  Tolerances as above: trajectory <= 1e-3, gradients <= 5e-3."""
    import ode_rl_amd
    from oracle import reference_modules as rm
    from oracle import torchdiffeq_ref
    f, _ = _f()
    with torch.no_grad():   # kink-free dynamics (see test_hip_backward.py)
        for i in (0, 2, 4, 6):
            f.gradient_net[i].weight.mul_(0.15)
            f.gradient_net[i].bias.copy_(torch.where(torch.arange(64) % 2 == 0, 2.5, -2.5))
        f.gradient_net[8].weight.mul_(4.0)
    sd = {k: v.detach().clone() for k, v in f.state_dict().items()}
    ws, bs = rm.split_convnet_state(sd, "gradient_net.")
    ws = [w.requires_grad_(True) for w in ws]
    bs = [b.requires_grad_(True) for b in bs]
    g = torch.Generator().manual_seed(40)
    z0 = (torch.randn(2, 64, 16, 16, generator=g) * 0.5).requires_grad_(True)
    t = torch.arange(20, 60, dtype=torch.float64) / 60      # 20 observed + 40 predicted frames
    gout = torch.randn(40, 2, 64, 16, 16, generator=g)
    ref = torchdiffeq_ref.odeint(rm.ode_func(ws, bs, compute_dtype="bf16"), z0, t, method="rk4")
    rg = torch.autograd.grad(ref, [z0] + ws + bs, gout)
    with torch.no_grad():
        exact = torchdiffeq_ref.odeint(rm.ode_func(ws, bs), z0, t, method="rk4")
    f = f.to(cuda)
    zd = z0.detach().to(cuda).requires_grad_(True)
    sol = ode_rl_amd.odeint(f, zd, t, method="rk4")
    assert sol.shape == (40, 2, 64, 16, 16)
    z = z0.detach()
    assert record("config4.traj.increment", rel_l2(sol.detach().cpu()[1:] - z, ref.detach()[1:] - z)) <= 1e-3
    e = record("config4.traj.increment.vs_fp32", rel_l2(sol.detach().cpu()[1:] - z, exact[1:] - z))
    assert 1e-4 <= e <= 2e-2, e     # really bf16, and of the expected size after 39 intervals
    sol.backward(gout.to(cuda))
    errs = [record("config4.grad_z0", rel_l2(zd.grad, rg[0]))]
    convs = [m for m in f.gradient_net if isinstance(m, torch.nn.Conv2d)]
    for i, (c, gw, gb) in enumerate(zip(convs, rg[1:6], rg[6:])):
        errs += [record(f"config4.grad_w{i}", rel_l2(c.weight.grad, gw)), record(f"config4.grad_b{i}", rel_l2(c.bias.grad, gb))]
    assert max(errs) <= 5e-3, errs


@pytest.mark.parametrize("method,batch,n_times", [("rk4", 64, 10), ("rk4", 3, 40), ("midpoint", 130, 5), ("euler", 7, 4), ("rk4", 128, 40)])
def test_whole_trajectory_launch_is_bit_identical_to_per_evaluation_launches(cuda, bf16_mode, method, batch, n_times):
    """bf16 inference: the whole trajectory in ONE launch (one workgroup per sample, state and stage derivatives in registers,
    activations in LDS; ftraj_bf16_kernel) against one fused launch per evaluation of f: same bf16 roundings and the same
    stage-combine expressions, so the trajectories must be equal bit for bit; and against the bf16-emulating oracle.
    ("rk4", 128, 40) is BASELINE configs[4]'s per-GPU size exactly as bench.py times it."""
    import ode_rl_amd
    from oracle import reference_modules as rm
    from oracle import torchdiffeq_ref
    lib = ode_rl_amd._lib.load()
    f, sd = _f(seed=3)
    f = f.to(cuda)
    z0 = torch.randn(batch, 64, 16, 16, generator=torch.Generator().manual_seed(batch)) * 0.5
    t = torch.arange(n_times, 2 * n_times, dtype=torch.float64) / (2 * n_times)
    was = lib.odehip_set_persistent_trajectory(0)
    try:
        with torch.no_grad():
            ref = ode_rl_amd.odeint(f, z0.to(cuda), t, method=method)
            lib.odehip_set_persistent_trajectory(1)
            n0 = lib.odehip_persistent_trajectory_launches()
            for _ in range(2):
                out = ode_rl_amd.odeint(f, z0.to(cuda), t, method=method)
        if os.environ.get("ODEHIP_PERSISTENT", "1") != "0":
            assert lib.odehip_persistent_trajectory_launches() == n0 + 2, "the whole-trajectory launch did not run"
        assert torch.equal(out, ref)
    finally:
        lib.odehip_set_persistent_trajectory(was)
    if batch <= 8:
        ws, bs = rm.split_convnet_state(sd, "gradient_net.")
        with torch.no_grad():
            emu = torchdiffeq_ref.odeint(rm.ode_func(ws, bs, compute_dtype="bf16"), z0, t, method=method)
        assert record(f"bf16.traj.{method}.T{n_times}.increment", rel_l2(out.cpu()[1:] - z0, emu[1:] - z0)) <= 1e-3


@pytest.mark.parametrize("batch,n_times", [(3, 4), (64, 10), (5, 40), (128, 40)])
def test_whole_trajectory_training_matches_per_evaluation_launches(cuda, bf16_mode, batch, n_times):
    """bf16 rk4 TRAINING step: saving forward + reverse sweep as one launch each (activations and conv-output gradients saved
    as bf16, gradient state in registers) + weight gradients on the bf16 operands, against the per-evaluation path (fp32 saves,
    bookkeeping in the conv epilogues).  Same bf16 roundings, same MFMA order, same bookkeeping expressions: the trajectory is
    equal bit for bit; the gradients are equal bit for bit on short grids and within 1e-5 on long ones (observed <= 1.3e-6: the
    compiler contracts a multiply-add of the interval-closing expression differently in the two kernels, and an ulp in front of a
    bf16 rounding flips it now and then); the bias gradients are summed in a different fixed order (observed 1e-7)."""
    import ode_rl_amd
    lib = ode_rl_amd._lib.load()
    f, _ = _f(seed=4)
    f = f.to(cuda)
    g = torch.Generator().manual_seed(100 + batch)
    z0 = (torch.randn(batch, 64, 16, 16, generator=g) * 0.5).to(cuda)
    t = torch.arange(n_times, 2 * n_times, dtype=torch.float64) / (2 * n_times)
    gout = torch.randn(n_times, batch, 64, 16, 16, generator=g).to(cuda)

    def run():
        f.zero_grad()
        z = z0.clone().requires_grad_(True)
        out = ode_rl_amd.odeint(f, z, t, method="rk4")
        out.backward(gout)
        convs = [m for m in f.gradient_net if isinstance(m, torch.nn.Conv2d)]
        return out.detach().clone(), z.grad.clone(), [c.weight.grad.clone() for c in convs], [c.bias.grad.clone() for c in convs]

    was = lib.odehip_set_persistent_trajectory(0)
    try:
        ref = run()
        lib.odehip_set_persistent_trajectory(1)
        n0 = lib.odehip_persistent_trajectory_launches()
        got = run()
        if os.environ.get("ODEHIP_PERSISTENT", "1") != "0":
            assert lib.odehip_persistent_trajectory_launches() == n0 + 1, "the whole-trajectory saving forward did not run"
        again = run()
    finally:
        lib.odehip_set_persistent_trajectory(was)
    assert torch.equal(got[0], ref[0])
    assert record(f"bf16.train.gz0.B{batch}", rel_l2(got[1], ref[1])) <= 1e-5
    for l, (a, b) in enumerate(zip(got[2], ref[2])):
        assert record(f"bf16.train.gw{l}.B{batch}", rel_l2(a, b)) <= 1e-5
    for l, (a, b) in enumerate(zip(got[3], ref[3])):
        assert record(f"bf16.train.gb{l}.B{batch}", rel_l2(a, b)) <= 1e-5
    for a, b in zip([again[0], again[1]] + again[2] + again[3], [got[0], got[1]] + got[2] + got[3]):
        assert torch.equal(a, b)     # deterministic
        assert bool(torch.isfinite(a).all())


@pytest.mark.parametrize("n_layers,batch,n_times", [(1, 2, 2), (5, 3, 3), (3, 1, 2)])
def test_whole_trajectory_launches_with_other_stack_depths(cuda, bf16_mode, n_layers, batch, n_times):
    """Edge cases of the one-launch bf16 paths: stacks of 3 and 7 convs (create_convnet n_layers 1 / 5), a single interval, a
    single sample -- forward bit-identical to per-evaluation launches, gradients within the bounds of the test above."""
    import ode_rl_amd
    lib = ode_rl_amd._lib.load()
    torch.manual_seed(20 + n_layers)
    f = ode_rl_amd.ODEFunc(64, 64, n_layers, 64, False, "relu", final_act=False).to(cuda)
    assert sum(isinstance(m, torch.nn.Conv2d) for m in f.gradient_net) == n_layers + 2
    g = torch.Generator().manual_seed(7 * n_layers + batch)
    z0 = (torch.randn(batch, 64, 16, 16, generator=g) * 0.5).to(cuda)
    t = torch.arange(n_times, 2 * n_times, dtype=torch.float64) / (2 * n_times)
    gout = torch.randn(n_times, batch, 64, 16, 16, generator=g).to(cuda)

    def run(train):
        if not train:
            with torch.no_grad():
                return [ode_rl_amd.odeint(f, z0, t, method="rk4")]
        f.zero_grad()
        z = z0.clone().requires_grad_(True)
        out = ode_rl_amd.odeint(f, z, t, method="rk4")
        out.backward(gout)
        return [out.detach().clone(), z.grad.clone()] + [p.grad.clone() for p in f.parameters()]

    was = lib.odehip_set_persistent_trajectory(0)
    try:
        ref_f, ref_t = run(False), run(True)
        lib.odehip_set_persistent_trajectory(1)
        n0 = lib.odehip_persistent_trajectory_launches()
        got_f, got_t = run(False), run(True)
        if os.environ.get("ODEHIP_PERSISTENT", "1") != "0":
            assert lib.odehip_persistent_trajectory_launches() == n0 + 2
    finally:
        lib.odehip_set_persistent_trajectory(was)
    assert torch.equal(got_f[0], ref_f[0]) and torch.equal(got_t[0], ref_t[0])
    for a, b in zip(got_t[1:], ref_t[1:]):
        assert rel_l2(a, b) <= 1e-5
