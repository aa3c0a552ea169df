"""ORACLE (test infrastructure only) -- CPU restatement of torchdiffeq==0.2.1 `odeint`.

This file is the checker for the HIP path, never the product: only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.

What it restates
----------------
The reference delegates its ODE arithmetic to the un-vendored third-party package
`torchdiffeq` (pinned `torchdiffeq==0.2.1` at /root/reference/requirements.txt:102; call
sites /root/reference/modules/DiffEqSolver.py:37,45-46).  The package is absent from
/root/reference and from this image, so the algorithm below is a restatement of the
*published* torchdiffeq 0.2.1 algorithm (`_impl/odeint.py`, `_impl/solvers.py`,
`_impl/fixed_grid.py`, `_impl/rk_common.py`, `_impl/dopri5.py`, `_impl/misc.py`,
`_impl/interp.py`, `_impl/adjoint.py`) as summarised in SURVEY.md section 3.3.

PARITY UNPINNED for the solver internals: the reference holds no golden vectors or tests at
the `odeint` boundary and torchdiffeq itself cannot be run here.  The
restatement is pinned instead by known-answer tests that need no reference
(tests/test_oracle_known_answers.py): scipy's RK45 tableau and the 2/3 error-weight
relation, the 3/8-rule stability polynomial, expm for linear systems, scipy solve_ivp.

Semantics kept (each is a torchdiffeq 0.2.1 behaviour the reference observes):
  * `solution[0] = y0`; T time points => T-1 integration intervals; len(t)==1 returns y0.
  * time is float64, state math is in y0.dtype; `dt` enters state arithmetic as a 0-dim
    tensor, i.e. rounded to y0.dtype.
  * `rk4` is the 3/8 rule (`rk4_alt_step_func`), one step per output interval.
  * dopri5: global RMS norm over the WHOLE tensor (batch included), FSAL, Shampine error
    weights, initial-step heuristic, quartic dense output; outputs are interpolated.
  * strictly decreasing `t` is handled by negating time and the dynamics.
"""
import math

import torch

# --------------------------------------------------------------------------- tableau
# Dormand-Prince-Shampine tableau (torchdiffeq/_impl/dopri5.py).
DP_ALPHA = [1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0]
DP_BETA = [
    [1 / 5],
    [3 / 40, 9 / 40],
    [44 / 45, -56 / 15, 32 / 9],
    [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
    [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
    [35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84],
]
DP_C_SOL = [35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0.0]
DP_C_ERROR = [
    35 / 384 - 1951 / 21600,
    0.0,
    500 / 1113 - 22642 / 50085,
    125 / 192 - 451 / 720,
    -2187 / 6784 - -12231 / 42400,
    11 / 84 - 649 / 6300,
    -1.0 / 60.0,
]
DP_C_MID = [
    6025192743 / 30085553152 / 2,
    0.0,
    51252292925 / 65400821598 / 2,
    -2691868925 / 45128329728 / 2,
    187940372067 / 1594534317056 / 2,
    -1776094331 / 19743644256 / 2,
    11237099 / 235043384 / 2,
]

SAFETY, IFACTOR, DFACTOR, ORDER = 0.9, 10.0, 0.2, 5
MAX_NUM_STEPS = 2 ** 31 - 1


def rms_norm(x):
    """torchdiffeq/_impl/misc.py:_rms_norm -- one scalar over the whole tensor."""
    return x.pow(2).mean().sqrt()


# --------------------------------------------------------------------------- fixed grid
def _euler_step(func, t0, dt, t1, y0):
    f0 = func(t0, y0)
    return dt * f0, f0


def _midpoint_step(func, t0, dt, t1, y0):
    half_dt = 0.5 * dt
    f0 = func(t0, y0)
    y_mid = y0 + f0 * half_dt
    return dt * func(t0 + half_dt, y_mid), f0


def _rk4_38_step(func, t0, dt, t1, y0):
    """torchdiffeq/_impl/rk_common.py:rk4_alt_step_func (the 3/8 rule)."""
    one_third, two_thirds = 1.0 / 3.0, 2.0 / 3.0
    k1 = func(t0, y0)
    k2 = func(t0 + dt * one_third, y0 + dt * k1 * one_third)
    k3 = func(t0 + dt * two_thirds, y0 + dt * (k2 - k1 * one_third))
    k4 = func(t1, y0 + dt * (k1 - k2 + k3))
    return (k1 + 3 * (k2 + k3) + k4) * dt * 0.125, k1


_FIXED = {"euler": _euler_step, "midpoint": _midpoint_step, "rk4": _rk4_38_step}


def _linear_interp(t0, t1, y0, y1, t):
    if t == t0:
        return y0
    if t == t1:
        return y1
    slope = (t - t0) / (t1 - t0)
    return y0 + slope.to(y0.dtype) * (y1 - y0)


def _integrate_fixed(func, y0, t, method, stats):
    """torchdiffeq/_impl/solvers.py:FixedGridODESolver.integrate, grid == t."""
    step = _FIXED[method]
    solution = torch.empty(len(t), *y0.shape, dtype=y0.dtype)
    solution[0] = y0
    j = 1
    for t0, t1 in zip(t[:-1], t[1:]):
        dt = t1 - t0
        dy, _ = step(func, t0, dt, t1, y0)
        y1 = y0 + dy
        while j < len(t) and t1 >= t[j]:
            solution[j] = _linear_interp(t0, t1, y0, y1, t[j])
            j += 1
        y0 = y1
    return solution


# --------------------------------------------------------------------------- dopri5
def _select_initial_step(func, t0, y0, order, rtol, atol, f0, norm):
    """torchdiffeq/_impl/misc.py:_select_initial_step (called with order = 5 - 1)."""
    dtype = y0.dtype
    t0f = t0.to(dtype)
    scale = atol + torch.abs(y0) * rtol
    d0 = norm(y0 / scale)
    d1 = norm(f0 / scale)
    if d0 < 1e-5 or d1 < 1e-5:
        h0 = torch.tensor(1e-6, dtype=dtype)
    else:
        h0 = 0.01 * d0 / d1
    y1 = y0 + h0 * f0
    f1 = func(t0f + h0, y1)
    d2 = norm((f1 - f0) / scale) / h0
    if d1 <= 1e-15 and d2 <= 1e-15:
        h1 = torch.max(torch.tensor(1e-6, dtype=dtype), h0 * 1e-3)
    else:
        h1 = (0.01 / max(d1, d2)) ** (1.0 / float(order + 1))
    return torch.min(100 * h0, h1).to(t0.dtype)


def _rk_step(func, y0, f0, t0, dt, t1):
    """torchdiffeq/_impl/rk_common.py:_runge_kutta_step for the dopri5 tableau.

    Returns y1, f1, y1_error and the 7 stage derivatives (a list, not a trailing axis).
    """
    dtype = y0.dtype
    t0s, dts, t1s = t0.to(dtype), dt.to(dtype), t1.to(dtype)
    k = [f0]
    yi = y0
    for alpha_i, beta_i in zip(DP_ALPHA, DP_BETA):
        ti = t1s if alpha_i == 1.0 else t0s + alpha_i * dts
        acc = torch.zeros_like(y0)
        for kj, bij in zip(k, beta_i):
            acc = acc + kj * (torch.tensor(bij, dtype=dtype) * dts)
        yi = y0 + acc
        k.append(func(ti, yi))
    y1 = yi  # c_sol == last beta row
    f1 = k[-1]
    err = torch.zeros_like(y0)
    for kj, cj in zip(k, DP_C_ERROR):
        err = err + kj * (dts * torch.tensor(cj, dtype=dtype))
    return y1, f1, err, k


def _error_ratio(err, rtol, atol, y0, y1, norm):
    tol = atol + rtol * torch.max(y0.abs(), y1.abs())
    return norm(err / tol)


def _optimal_step_size(last_step, error_ratio):
    """torchdiffeq/_impl/misc.py:_optimal_step_size (order 5).  Decorated @torch.no_grad() there: the next step size is a
    constant of the autograd graph (only the very first dt, from _select_initial_step, carries a gradient)."""
    last_step = last_step.detach()
    if error_ratio == 0:
        return last_step * IFACTOR
    dfactor = 1.0 if error_ratio < 1 else DFACTOR
    er = error_ratio.to(last_step.dtype)
    exponent = 1.0 / ORDER
    factor = min(IFACTOR, max(SAFETY / float(er ** exponent), dfactor))
    return last_step * factor


def _interp_fit(y0, y1, k, dt):
    dts = dt.to(y0.dtype)
    y_mid = y0.clone()
    for kj, cj in zip(k, DP_C_MID):
        y_mid = y_mid + kj * (dts * torch.tensor(cj, dtype=y0.dtype))
    f0, f1 = k[0], k[-1]
    a = 2 * dts * (f1 - f0) - 8 * (y1 + y0) + 16 * y_mid
    b = dts * (5 * f0 - 3 * f1) + 18 * y0 + 14 * y1 - 32 * y_mid
    c = dts * (f1 - 4 * f0) - 11 * y0 - 5 * y1 + 16 * y_mid
    d = dts * f0
    e = y0
    return [e, d, c, b, a]


def _interp_evaluate(coeffs, t0, t1, t):
    x = ((t - t0) / (t1 - t0)).to(coeffs[0].dtype)
    total = coeffs[0] + x * coeffs[1]
    x_power = x
    for c in coeffs[2:]:
        x_power = x_power * x
        total = total + x_power * c
    return total


def _integrate_dopri5(func, y0, t, rtol, atol, stats, norm, options=None):
    solution = torch.empty(len(t), *y0.shape, dtype=y0.dtype)
    solution[0] = y0
    t = t.to(torch.float64)
    f0 = func(t[0].to(y0.dtype), y0)
    options = options or {}
    max_num_steps = options.get("max_num_steps", MAX_NUM_STEPS)
    if options.get("first_step") is None:
        dt = _select_initial_step(func, t[0], y0, ORDER - 1, rtol, atol, f0, norm)
        stats["nfe"] = stats.get("nfe", 0) + 2
    else:  # torchdiffeq options={'first_step': dt}: the heuristic (and its extra f-eval) is skipped
        dt = torch.as_tensor(options["first_step"], dtype=torch.float64)
        stats["nfe"] = stats.get("nfe", 0) + 1
    y, f, t0, t1 = y0, f0, t[0], t[0]
    interp = [y0] * 5
    for i in range(1, len(t)):
        next_t = t[i]
        n_steps = 0   # torchdiffeq 0.2.1 solvers.py `_advance(next_t)`: the counter is a LOCAL of the call -- max_num_steps bounds the
        #               attempted steps (accepted or rejected) spent on ONE output time, not the whole integration
        while next_t > t1:
            assert n_steps < max_num_steps, "max_num_steps exceeded ({}>={})".format(n_steps, max_num_steps)
            ta = t1
            tb = ta + dt
            assert tb > ta, "underflow in dt {}".format(dt.item())
            assert torch.isfinite(y).all(), "non-finite values in state `y`"
            y1, f1, err, k = _rk_step(func, y, f, ta, dt, tb)
            ratio = _error_ratio(err, rtol, atol, y, y1, norm)
            accept = bool(ratio <= 1)
            stats["nfe"] += 6
            stats.setdefault("dts", []).append(float(dt))
            stats.setdefault("ratios", []).append(float(ratio))
            if accept:
                interp = _interp_fit(y, y1, k, dt)
                t0, t1 = ta, tb
                y, f = y1, f1
                stats["n_accept"] = stats.get("n_accept", 0) + 1
            else:
                stats["n_reject"] = stats.get("n_reject", 0) + 1
            dt = _optimal_step_size(dt, ratio)
            n_steps += 1
        stats.setdefault("steps_per_output", []).append(n_steps)   # (test instrumentation) attempts this `_advance` call took
        solution[i] = _interp_evaluate(interp, t0, t1, next_t)
    return solution


# --------------------------------------------------------------------------- front door
def _check_t(t):
    assert t.ndim == 1, "`t` must be one dimensional"
    if len(t) > 1:
        d = t[1:] - t[:-1]
        assert bool((d > 0).all()) or bool((d < 0).all()), "t must be strictly increasing or decreasing"


def odeint(func, y0, t, rtol=1e-7, atol=1e-9, method=None, options=None, stats=None, norm=None):
    """`torchdiffeq.odeint(func, y0, t, rtol=, atol=, method=)` for a single-tensor state.

    `stats`, if given, is filled with nfe / n_accept / n_reject / dts (not a torchdiffeq
    argument; test instrumentation).
    """
    if method is None:
        method = "dopri5"
    if stats is None:
        stats = {}
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(t)
    _check_t(t)
    if len(t) > 1 and bool(t[0] > t[1]):
        t = -t
        fwd = func
        func = lambda tt, yy: -fwd(-tt, yy)  # noqa: E731  (_ReverseFunc)
    if method in _FIXED:
        per_step = {"euler": 1, "midpoint": 2, "rk4": 4}[method]
        stats["nfe"] = stats.get("nfe", 0) + per_step * (len(t) - 1)
        return _integrate_fixed(func, y0, t, method, stats)
    if method == "dopri5":
        return _integrate_dopri5(func, y0, t, rtol, atol, stats, norm or rms_norm, options)
    raise ValueError('Invalid method "{}".'.format(method))


# --------------------------------------------------------------------------- adjoint
def odeint_adjoint(func, y0, t, params, grad_out, rtol=1e-7, atol=1e-9, method=None,
                   stats=None, adjoint_norm="mixed"):
    """torchdiffeq/_impl/adjoint.py semantics, returned functionally.

    Forward under no_grad, then the augmented system (y, a_y, a_theta) is integrated
    backwards interval by interval with the same method/tolerances; `y` is NOT reseeded
    from the stored trajectory inside an interval but restarts from `ys[i-1]` at each
    output time; `a_y += grad_out[i-1]` there.  For adaptive methods the error norm of the
    tuple state is the max over per-tensor RMS norms (mixed norm of torchdiffeq 0.2.1), which
    here is realised by integrating the flat state with that norm.  adjoint_norm="seminorm"
    (adjoint_options={"norm": "seminorm"}, adjoint.py handle_adjoint_norm_) leaves the parameter
    block out of the error norm: max(rms(y), rms(a_y)).

    Returns (ys, grad_y0, [grad_theta...]).
    """
    params = list(params)
    with torch.no_grad():
        ys = odeint(func, y0, t, rtol=rtol, atol=atol, method=method)
    if method is None:
        method = "dopri5"
    n_y = y0.numel()
    shapes = [p.shape for p in params]
    sizes = [p.numel() for p in params]

    def aug_dynamics(tt, state):
        y = state[:n_y].view_as(y0)
        a_y = state[n_y:2 * n_y].view_as(y0)
        with torch.enable_grad():
            yr = y.detach().requires_grad_(True)
            fy = func(tt, yr)
            grads = torch.autograd.grad(fy, [yr] + params, -a_y, allow_unused=True)
        vjp_y = grads[0] if grads[0] is not None else torch.zeros_like(y)
        vjp_p = [g if g is not None else torch.zeros_like(p) for g, p in zip(grads[1:], params)]
        return torch.cat([fy.detach().reshape(-1), vjp_y.reshape(-1)] + [g.reshape(-1) for g in vjp_p])

    a_y = grad_out[-1].clone()
    a_p = torch.zeros(sum(sizes), dtype=y0.dtype)
    if adjoint_norm not in ("mixed", "seminorm"):
        raise ValueError("adjoint_norm must be 'mixed' or 'seminorm'")
    norm_chunks = [n_y, n_y] + (sizes if adjoint_norm == "mixed" else [])

    def mixed_norm(x):
        out, off = None, 0
        for n in norm_chunks:
            v = rms_norm(x[off:off + n])
            out = v if out is None else torch.max(out, v)
            off += n
        return out

    with torch.no_grad():
        for i in range(len(t) - 1, 0, -1):
            state = torch.cat([ys[i].reshape(-1), a_y.reshape(-1), a_p])
            tt = torch.stack([t[i], t[i - 1]])
            sol = odeint(aug_dynamics, state, tt, rtol=rtol, atol=atol, method=method, stats=stats,
                         norm=mixed_norm)
            end = sol[1]
            a_y = end[n_y:2 * n_y].view_as(y0) + grad_out[i - 1]
            a_p = end[2 * n_y:]
    grads_p, off = [], 0
    for s, n in zip(shapes, sizes):
        grads_p.append(a_p[off:off + n].view(s).clone())
        off += n
    return ys, a_y, grads_p
