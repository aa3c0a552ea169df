"""Error behaviour of the training-side entry points (C ABI status codes surface as Python exceptions; nothing falls back).
Shapes the HIP path does not serve must be refused loudly, and calls that would overrun a workspace must be rejected before
any kernel is launched."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_backward_refuses_unsupported_dynamics(cuda):
    """Backward needs channel counts that are multiples of 64: a 32-channel f integrates forward but must not pretend to
    differentiate."""
    import ode_rl_amd
    f = ode_rl_amd.ODEFunc(32, 32, 3, 32, False, "relu", final_act=False).to(cuda)
    z0 = torch.randn(2, 32, 16, 16, device=cuda, requires_grad=True)
    t = torch.tensor([0.0, 0.5, 1.0], dtype=torch.float64)
    with torch.no_grad():
        assert ode_rl_amd.odeint(f, z0, t, method="rk4").shape == (3, 2, 32, 16, 16)
    for method in ("rk4", "dopri5"):
        with pytest.raises(ValueError):
            ode_rl_amd.odeint(f, z0, t, method=method).sum().backward()
    with pytest.raises(ValueError):
        ode_rl_amd.odeint_adjoint(f, z0, t, method="rk4").sum().backward()


def test_adjoint_option_errors(cuda):
    import ode_rl_amd
    f = ode_rl_amd.ODEFunc(64, 64, 3, 64, False, "relu", final_act=False).to(cuda)
    z0 = torch.randn(1, 64, 16, 16, device=cuda, requires_grad=True)
    t = torch.tensor([0.0, 0.5], dtype=torch.float64)
    with pytest.raises(ValueError):               # unknown adjoint norm
        ode_rl_amd.odeint_adjoint(f, z0, t, method="dopri5", adjoint_options={"norm": "bogus"})
    with pytest.raises(ValueError):
        ode_rl_amd.odeint_adjoint(f, z0, t, method="dopri5", adjoint_options={"norm": "seminorm", "bogus": 1})
    with pytest.raises(ValueError):
        ode_rl_amd.odeint_adjoint(f, z0, t, method="adams")
    with pytest.raises(NotImplementedError):      # decreasing t under autograd
        ode_rl_amd.odeint(f, z0, torch.tensor([1.0, 0.5], dtype=torch.float64), method="rk4")
    # too few slots for the accepted backward steps: reported, not overrun
    sol = ode_rl_amd.odeint_adjoint(f, z0, torch.tensor([0.0, 0.5, 1.0], dtype=torch.float64), method="dopri5",
                                    adjoint_options={"norm": "seminorm", "max_accept": 1})
    with pytest.raises(ValueError):
        sol.sum().backward()


def test_c_abi_rejects_small_workspaces_and_bad_arguments(cuda):
    from ode_rl_amd import _lib, hip_ops
    from ode_rl_amd.odeint import conv_stack_of
    import ode_rl_amd
    lib = _lib.load()
    f = ode_rl_amd.ODEFunc(64, 64, 3, 64, False, "relu", final_act=False).to(cuda)
    stack = conv_stack_of(f)
    desc, dg = stack.refresh(), stack.dgrad_desc()
    b, n = 2, 3
    z0 = torch.randn(b, 64, 16, 16, device=cuda)
    out = torch.empty(n, b, 64, 16, 16, device=cuda)
    t = (ctypes.c_double * n)(0.0, 0.5, 1.0)
    need = lib.odehip_odeint_workspace_bytes(ctypes.byref(desc), b, n, 2, 1)
    ws = torch.empty(need // 2, dtype=torch.uint8, device=cuda)
    rc = lib.odehip_odeint_fixed(ctypes.byref(desc), 2, z0.data_ptr(), t, n, b, out.data_ptr(), 1, 0, ws.data_ptr(), ws.numel(), ctypes.byref(ctypes.c_int(0)), None)
    assert rc == -1 and b"workspace too small" in lib.odehip_last_error()
    # backward through dopri5 with a step log that does not tile [t0, t1]
    gw = (ctypes.c_void_p * 5)(*[torch.empty_like(c.weight).data_ptr() for c in stack.convs])
    gb = (ctypes.c_void_p * 5)(*[torch.empty_like(c.bias).data_ptr() for c in stack.convs])
    log = (ctypes.c_double * 4)(0.0, 0.3, 0.4, 0.6)          # gap between 0.3 and 0.4
    need = lib.odehip_dopri5_backward_workspace_bytes(ctypes.byref(desc), b, n, 2)
    ws = torch.empty(need, dtype=torch.uint8, device=cuda)
    rc = lib.odehip_odeint_dopri5_backward(ctypes.byref(desc), ctypes.byref(dg), t, n, b, log, 2, z0.data_ptr(), out.data_ptr(),
                                           z0.data_ptr(), gw, gb, ws.data_ptr(), ws.numel(), None)
    assert rc == -1 and b"not contiguous" in lib.odehip_last_error()
    # Adam: steps count from 1
    p = torch.zeros(8, device=cuda)
    arr = (ctypes.c_void_p * 1)(p.data_ptr())
    rc = lib.odehip_adam_step(arr, arr, arr, arr, (ctypes.c_longlong * 1)(8), 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, None)
    assert rc == -1
    hip_ops.check_canaries()


def _fault_run(which):
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "_fault_worker.py"), which], capture_output=True, text=True, timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("FAULT_RECORD ")]
    assert lines, (r.returncode, r.stdout[-800:], r.stderr[-1500:])
    return json.loads(lines[-1][len("FAULT_RECORD "):])


@pytest.mark.gpu
def test_a_lost_partner_in_a_trajectory_walk_is_loud(cuda):
    """Fault injection (ODEHIP_FAULT_INJECT=1, a subprocess: the give-up disables persistent launches for its process): logical
    workgroup 0 of the sixteen-workgroup walk leaves in front of row 1 of a whole-trajectory launch.  Its partners' capped waits give
    up instead of hanging the device; the caller gets NaN frames (never plausible numbers), the next library call raises, and the
    library carries on with one launch per layer -- bit-identical to what the walk would have produced."""
    import os
    if os.environ.get("ODEHIP_PERSISTENT") == "0" or os.environ.get("ODEHIP_PERSIST16") == "0":
        pytest.skip("the sixteen-workgroup walk is switched off for this run")
    rec = _fault_run("trajectory")
    if not rec.get("raised_in_call"):
        assert rec["later_frames_all_nan"], rec   # (the guard fills the whole result, solution[0] included)
        assert rec["next_call_raised"], rec
    assert rec["usable_afterwards"], rec


@pytest.mark.gpu
def test_a_lost_partner_in_a_single_evaluation_walk_is_loud(cuda):
    """The same fault in the encoder loop at batch 2, whose Euler steps run as single-evaluation walks since round 4 (batches up to 16;
    VERDICT r03: "no NaN guard -- keep it off"): there is no guard launch behind those, so the walk itself NaN-fills its outputs when a
    wait of the launch gave up (nan_fill_row16).  The encoder's hidden states carry NaN (the lost workgroup's partners wrote them; they
    are what is checked: when this test was written forward()'s (mean, std) sat behind ReLUs that mapped NaN to 0 -- the finding
    behind test_a_non_finite_state_is_not_laundered), the next call raises, the library is usable afterwards."""
    import os
    if os.environ.get("ODEHIP_PERSISTENT") == "0" or os.environ.get("ODEHIP_PERSIST16") == "0" or os.environ.get("ODEHIP_EVAL_WALK") == "0":
        pytest.skip("single-evaluation walks are switched off for this run")
    rec = _fault_run("encoder")
    if not rec.get("raised_in_call"):
        assert rec["output_has_nan"] and not rec["output_equals_good"], rec
        assert rec["next_call_raised"], rec
    assert rec["usable_afterwards"], rec


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["rk4", "euler"])
def test_a_non_finite_state_is_not_laundered(cuda, method):
    """torch.relu propagates NaN; v_max_f32(x, 0) returns 0 for it, which would turn a non-finite activation into a plausible zero
    (round 4: every ReLU of the library is `x < 0 ? 0 : x`).  A NaN in one sample's z0: wherever the oracle's trajectory is NaN the HIP
    trajectory is NaN too (Winograd tiles may spread it further, never less); the other sample of the batch is untouched, bit for bit."""
    import ode_rl_amd
    from oracle import reference_modules as rm
    from oracle import torchdiffeq_ref
    torch.manual_seed(2)
    f = ode_rl_amd.ODEFunc(64, 64, 3, 64, False, "relu", final_act=False)
    ws, bs = rm.split_convnet_state({k: v.detach().clone() for k, v in f.state_dict().items()}, "gradient_net.")
    z0 = torch.randn(2, 64, 16, 16) * 0.5
    clean = z0.clone()
    z0[0, 5, 7, 9] = float("nan")
    t = torch.tensor([0.0, 0.3, 0.5], dtype=torch.float64)
    with torch.no_grad():
        ref = torchdiffeq_ref.odeint(rm.ode_func(ws, bs), z0, t, method=method)
        fd = f.to(cuda)
        out = ode_rl_amd.odeint(fd, z0.to(cuda), t, method=method).cpu()
        base = ode_rl_amd.odeint(fd, clean.to(cuda), t, method=method).cpu()
    assert bool(torch.isnan(ref[1:, 0]).any())
    assert int((torch.isnan(ref) & ~torch.isnan(out)).sum()) == 0          # nothing the reference semantics keep NaN became a number
    assert bool(torch.isfinite(out[:, 1]).all()) and torch.equal(out[:, 1], base[:, 1])
