"""Known-answer tests that pin the torchdiffeq-0.2.1 restatement (oracle/torchdiffeq_ref.py) WITHOUT the
reference: the package is un-vendored and not installed, so these are what stands between a mis-recalled
coefficient and silent drift (section 8c F8)."""
import math

import numpy as np
import pytest
import scipy.integrate
import scipy.linalg
import torch

from oracle import torchdiffeq_ref as td


def test_tableau_matches_scipy_rk45():
    rk = scipy.integrate.RK45
    np.testing.assert_allclose(td.DP_ALPHA, list(rk.C[1:]) + [1.0], rtol=0, atol=1e-15)
    for i, row in enumerate(td.DP_BETA[:-1]):
        np.testing.assert_allclose(row, rk.A[i + 1][:i + 1], rtol=1e-15)
    np.testing.assert_allclose(td.DP_BETA[-1], rk.B, rtol=1e-15)
    np.testing.assert_allclose(td.DP_C_SOL[:-1], rk.B, rtol=1e-15)
    assert td.DP_C_SOL[-1] == 0.0
    # Shampine's scaled error weights: c_error = 2/3 * (-E) in all 7 entries
    np.testing.assert_allclose(td.DP_C_ERROR, (-2.0 / 3.0) * np.asarray(rk.E), rtol=1e-12, atol=1e-17)
    # rows of the tableau sum to the node
    for a, row in zip(td.DP_ALPHA, td.DP_BETA):
        assert abs(sum(row) - a) < 1e-15
    # the mid-point weights sum to 1/2
    assert abs(sum(td.DP_C_MID) - 0.5) < 1e-15


def _linear(A):
    return lambda t, y: y @ A.T


def test_rk4_is_the_three_eighths_rule_not_classic():
    # y' = y^2 (non-linear, scalar) separates the 3/8 rule from classic RK4 at O(h^5)
    f = lambda t, y: y * y
    y0 = torch.tensor([1.0], dtype=torch.float64)
    h = 0.1
    t = torch.tensor([0.0, h], dtype=torch.float64)
    got = float(td.odeint(f, y0, t, method="rk4")[1])

    def step(y, kind):
        if kind == "38":
            k1 = y * y
            k2 = (y + h * k1 / 3) ** 2
            k3 = (y + h * (k2 - k1 / 3)) ** 2
            k4 = (y + h * (k1 - k2 + k3)) ** 2
            return y + h * (k1 + 3 * (k2 + k3) + k4) / 8
        k1 = y * y
        k2 = (y + h * k1 / 2) ** 2
        k3 = (y + h * k2 / 2) ** 2
        k4 = (y + h * k3) ** 2
        return y + h * (k1 + 2 * k2 + 2 * k3 + k4) / 6
    assert abs(got - step(1.0, "38")) < 1e-15
    assert abs(got - step(1.0, "classic")) > 1e-9


def test_fixed_grid_stability_polynomials():
    # one step on y' = A y equals R(hA) y with R the method's stability polynomial
    A = torch.tensor([[-0.3, 1.1], [-0.7, -0.2]], dtype=torch.float64)
    y0 = torch.tensor([[0.4, -1.3]], dtype=torch.float64)
    h = 0.25
    t = torch.tensor([1.0, 1.0 + h], dtype=torch.float64)
    Z = h * A
    I = torch.eye(2, dtype=torch.float64)
    R = {"euler": I + Z, "midpoint": I + Z + Z @ Z / 2,
         "rk4": I + Z + Z @ Z / 2 + Z @ Z @ Z / 6 + Z @ Z @ Z @ Z / 24}
    for m, Rm in R.items():
        got = td.odeint(_linear(A), y0, t, method=m)[1]
        torch.testing.assert_close(got, y0 @ Rm.T, rtol=1e-13, atol=1e-15)


def test_solution0_is_y0_and_shapes():
    A = torch.tensor([[-0.5]], dtype=torch.float64)
    y0 = torch.randn(3, 1, dtype=torch.float64)
    for m in ("euler", "midpoint", "rk4", "dopri5"):
        sol = td.odeint(_linear(A), y0, torch.tensor([0.0, 0.3, 0.9], dtype=torch.float64), method=m)
        assert sol.shape == (3, 3, 1) and torch.equal(sol[0], y0)
        one = td.odeint(_linear(A), y0, torch.tensor([0.7], dtype=torch.float64), method=m)
        assert one.shape == (1, 3, 1) and torch.equal(one[0], y0)
    with pytest.raises(ValueError):
        td.odeint(_linear(A), y0, torch.tensor([0.0, 1.0]), method="adams")
    with pytest.raises(AssertionError):
        td.odeint(_linear(A), y0, torch.tensor([0.0, 1.0, 0.5]), method="rk4")


def test_dopri5_matches_expm_and_scipy():
    A = torch.tensor([[-0.5, 2.0], [-2.0, -0.5]], dtype=torch.float64)
    y0 = torch.tensor([[1.0, 0.5]], dtype=torch.float64)
    t = torch.linspace(0.0, 2.0, 7, dtype=torch.float64)
    st = {}
    sol = td.odeint(_linear(A), y0, t, rtol=1e-9, atol=1e-11, method="dopri5", stats=st)
    for i, ti in enumerate(t):
        exact = torch.from_numpy(scipy.linalg.expm(A.numpy() * float(ti))) @ y0[0]
        torch.testing.assert_close(sol[i, 0], exact, rtol=1e-7, atol=1e-9)
    assert st["nfe"] == 2 + 6 * (st.get("n_accept", 0) + st.get("n_reject", 0))
    # non-linear: van der Pol against scipy at tight tolerance
    mu = 1.5
    f = lambda t, y: torch.stack([y[1], mu * (1 - y[0] ** 2) * y[1] - y[0]])
    ref = scipy.integrate.solve_ivp(lambda t, y: [y[1], mu * (1 - y[0] ** 2) * y[1] - y[0]], (0, 3), [2.0, 0.0],
                                    rtol=1e-11, atol=1e-13, t_eval=[1.0, 2.0, 3.0]).y.T
    sol = td.odeint(f, torch.tensor([2.0, 0.0], dtype=torch.float64), torch.tensor([0.0, 1.0, 2.0, 3.0], dtype=torch.float64),
                    rtol=1e-8, atol=1e-10, method="dopri5")
    np.testing.assert_allclose(sol[1:].numpy(), ref, rtol=2e-6, atol=2e-7)


def test_dopri5_controller_pieces():
    one = torch.tensor(0.1, dtype=torch.float64)
    assert float(td._optimal_step_size(one, torch.tensor(0.0))) == pytest.approx(1.0)           # ratio 0 -> x10
    assert float(td._optimal_step_size(one, torch.tensor(1e-12))) == pytest.approx(1.0)        # capped at ifactor
    assert float(td._optimal_step_size(one, torch.tensor(1.0))) == pytest.approx(0.09)         # safety
    assert float(td._optimal_step_size(one, torch.tensor(1e12))) == pytest.approx(0.02)        # floored at dfactor
    assert float(td._optimal_step_size(one, torch.tensor(0.5))) == pytest.approx(0.1 * 0.9 / 0.5 ** 0.2)
    # dense output: the quartic reproduces both ends and the mid-point formula
    f = lambda t, y: -y
    y0 = torch.tensor([1.0], dtype=torch.float64)
    t0, dt = torch.tensor(0.0, dtype=torch.float64), torch.tensor(0.2, dtype=torch.float64)
    y1, f1, err, k = td._rk_step(f, y0, f(t0, y0), t0, dt, t0 + dt)
    co = td._interp_fit(y0, y1, k, dt)
    torch.testing.assert_close(td._interp_evaluate(co, t0, t0 + dt, t0), y0)
    torch.testing.assert_close(td._interp_evaluate(co, t0, t0 + dt, t0 + dt), y1, rtol=1e-13, atol=0)
    mid = float(td._interp_evaluate(co, t0, t0 + dt, t0 + dt / 2))
    assert abs(mid - math.exp(-0.1)) < 5e-7  # 4th-order dense output, h = 0.2


def test_initial_step_is_hairers_algorithm_as_scipy_implements_it():
    """torchdiffeq's `_select_initial_step` is the starting-step heuristic of Hairer, Norsett & Wanner (II.4) with the RMS norm, called with
    order = 4 for dopri5 -- the algorithm scipy's RK45 uses (`scipy.integrate._ivp.common.select_initial_step`, same constants: 0.01,
    1e-5, 1e-6, 100 h0, exponent 1 / (order + 1)).  An independent implementation of the published algorithm: the restatement must
    reproduce its value (scipy's two extra bounds -- interval length, max_step -- are kept out of reach)."""
    from scipy.integrate._ivp.common import select_initial_step
    rms = lambda x: x.pow(2).mean().sqrt()
    cases = [
        (lambda t, y: np.array([y[1], 1.5 * (1 - y[0] ** 2) * y[1] - y[0]]), [2.0, 0.3], 1e-6, 1e-9),       # van der Pol
        (lambda t, y: np.array([-0.5 * y[0] + 2.0 * y[1], -2.0 * y[0] - 0.5 * y[1], np.sin(t) - y[2]]), [1.0, 0.5, -2.0], 1e-4, 1e-5),
        (lambda t, y: np.array([1e-7 * y[0]]), [1e-9], 1e-3, 1e-3),                                          # d0 < 1e-5: h0 = 1e-6
    ]
    for fun, y0, rtol, atol in cases:
        y0n = np.array(y0, dtype=np.float64)
        want = select_initial_step(fun, 0.25, y0n, 1e9, np.inf, fun(0.25, y0n), 1.0, 4, rtol, atol)
        ft = lambda t, y: torch.from_numpy(np.asarray(fun(float(t), y.numpy())))
        t0, y0t = torch.tensor(0.25, dtype=torch.float64), torch.from_numpy(y0n)
        got = td._select_initial_step(ft, t0, y0t, 4, rtol, atol, ft(t0, y0t), rms)
        assert float(got) == pytest.approx(float(want), rel=1e-12)


def test_error_ratio_is_the_scaled_rms_norm_scipy_uses():
    """One dopri5 step: the restatement's error ratio = RMS(err / (atol + rtol max(|y0|, |y1|))) is 2/3 of scipy RK45's
    `_estimate_error_norm` for the same step: same tableau (the step result agrees to round-off), same scale and norm, and torchdiffeq's
    error weights are Shampine's, 2/3 of the Dormand-Prince ones scipy uses (test_tableau_matches_scipy_rk45)."""
    from scipy.integrate import RK45
    fun = lambda t, y: np.array([y[1], 1.5 * (1 - y[0] ** 2) * y[1] - y[0]])
    y0n, h, rtol, atol = np.array([2.0, 0.3]), 0.05, 1e-6, 1e-9
    solver = RK45(fun, 0.0, y0n, 10.0, rtol=rtol, atol=atol, first_step=h)
    from scipy.integrate._ivp.rk import rk_step
    y_new, f_new = rk_step(fun, 0.0, y0n, fun(0.0, y0n), h, solver.A, solver.B, solver.C, solver.K)
    scale = atol + np.maximum(np.abs(y0n), np.abs(y_new)) * rtol
    want = solver._estimate_error_norm(solver.K, h, scale)
    ft = lambda t, y: torch.from_numpy(np.asarray(fun(float(t), y.numpy())))
    y0t, t0, dt = torch.from_numpy(y0n), torch.tensor(0.0, dtype=torch.float64), torch.tensor(h, dtype=torch.float64)
    y1, f1, err, k = td._rk_step(ft, y0t, ft(t0, y0t), t0, dt, t0 + dt)
    np.testing.assert_allclose(y1.numpy(), y_new, rtol=1e-13, atol=0)
    got = td._error_ratio(err, rtol, atol, y0t, y1, lambda x: x.pow(2).mean().sqrt())
    assert float(got) == pytest.approx(2.0 / 3.0 * float(want), rel=1e-9)


def test_order_conditions_on_polynomial_solutions():
    """Known answers that pin every coefficient at once: on dy/dt = 4 t^3 (solution t^4) a 5(4) pair is exact in both solutions, so the
    error estimate vanishes, and torchdiffeq's quartic dense output (the mid-point weights DP_C_MID + `_interp_fit`) reproduces the
    solution at ANY point of the step; on dy/dt = 5 t^4 the 5th-order solution is still exact and the 4th-order one is not."""
    t0, dt = torch.tensor(0.3, dtype=torch.float64), torch.tensor(0.5, dtype=torch.float64)
    y0 = torch.tensor([0.7], dtype=torch.float64)
    f4 = lambda t, y: (4 * t ** 3).reshape(1).to(torch.float64)
    y1, f1, err, k = td._rk_step(f4, y0, f4(t0, y0), t0, dt, t0 + dt)
    assert abs(float(y1) - (0.7 + 0.8 ** 4 - 0.3 ** 4)) < 1e-14 and abs(float(err)) < 1e-15
    co = td._interp_fit(y0, y1, k, dt)
    for x in (0.3, 0.35, 0.5, 0.61, 0.8):
        got = float(td._interp_evaluate(co, t0, t0 + dt, torch.tensor(x, dtype=torch.float64)))
        assert abs(got - (0.7 + x ** 4 - 0.3 ** 4)) < 1e-14, x
    f5 = lambda t, y: (5 * t ** 4).reshape(1).to(torch.float64)
    y1, f1, err, k = td._rk_step(f5, y0, f5(t0, y0), t0, dt, t0 + dt)
    assert abs(float(y1) - (0.7 + 0.8 ** 5 - 0.3 ** 5)) < 1e-14 and abs(float(err)) > 1e-6


def test_reversed_time_negates_dynamics():
    A = torch.tensor([[-0.5, 1.0], [-1.0, -0.5]], dtype=torch.float64)
    y0 = torch.tensor([[1.0, 0.0]], dtype=torch.float64)
    fwd = td.odeint(_linear(A), y0, torch.tensor([0.0, 0.5, 1.0], dtype=torch.float64), rtol=1e-10, atol=1e-12, method="dopri5")
    back = td.odeint(_linear(A), fwd[-1], torch.tensor([1.0, 0.5, 0.0], dtype=torch.float64), rtol=1e-10, atol=1e-12,
                     method="dopri5")
    torch.testing.assert_close(back[-1], y0, rtol=1e-7, atol=1e-9)
    torch.testing.assert_close(back[1], fwd[1], rtol=1e-7, atol=1e-9)


def test_adjoint_matches_autograd_through_the_solver():
    torch.manual_seed(0)
    lin = torch.nn.Linear(3, 3).double()
    f = lambda t, y: torch.tanh(lin(y))
    y0 = torch.randn(2, 3, dtype=torch.float64)
    t = torch.tensor([0.0, 0.4, 1.0], dtype=torch.float64)
    gout = torch.randn(3, 2, 3, dtype=torch.float64)
    # discretise-then-optimise gradient (what the reference does: autograd through the solver ops)
    y0r = y0.clone().requires_grad_(True)
    sol = td.odeint(f, y0r, t, rtol=1e-10, atol=1e-12, method="dopri5")
    g = torch.autograd.grad(sol, [y0r] + list(lin.parameters()), gout)
    ys, gy0, gp = td.odeint_adjoint(f, y0, t, list(lin.parameters()), gout, rtol=1e-10, atol=1e-12, method="dopri5")
    torch.testing.assert_close(ys, sol.detach(), rtol=1e-9, atol=1e-11)
    torch.testing.assert_close(gy0, g[0], rtol=1e-6, atol=1e-8)
    for a, b in zip(gp, g[1:]):
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-8)


def test_max_num_steps_bounds_the_steps_of_one_output_time_not_of_the_integration():
    """torchdiffeq 0.2.1 `AdaptiveStepsizeODESolver`-style `_advance(next_t)` (solvers.py / rk_common.py): `n_steps = 0` is a local
    of the call and `assert n_steps < self.max_num_steps` guards each step of THAT call ("one deviation").  A grid whose
    outputs each need a few steps tells the two readings apart: the whole integration takes more attempts than any one output."""
    f = lambda t, y: -3.0 * y * (1.0 + 0.5 * torch.sin(7.0 * t))
    y0 = torch.tensor([1.0, 2.0], dtype=torch.float64)
    t = torch.linspace(0.0, 4.0, 9, dtype=torch.float64)
    st = {}
    ref = td.odeint(f, y0, t, rtol=1e-7, atol=1e-9, method="dopri5", stats=st)
    per_out = st["steps_per_output"]
    assert len(per_out) == 8 and sum(per_out) == st["n_accept"] + st.get("n_reject", 0)
    worst = max(per_out)
    assert worst >= 2 and sum(per_out) > worst + 2, per_out        # the input separates the two semantics
    # bound == the worst single output: passes under torchdiffeq's per-call counter (a whole-integration counter would assert)
    got = td.odeint(f, y0, t, rtol=1e-7, atol=1e-9, method="dopri5", options={"max_num_steps": worst})
    assert torch.equal(got, ref)
    with pytest.raises(AssertionError, match="max_num_steps exceeded"):
        td.odeint(f, y0, t, rtol=1e-7, atol=1e-9, method="dopri5", options={"max_num_steps": worst - 1})
