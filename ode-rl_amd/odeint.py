"""`odeint(func, y0, t, rtol=, atol=, method=)` -- the call surface of torchdiffeq that the reference
consumes (/root/reference/modules/DiffEqSolver.py:37,45-46), executed by the HIP library.

Supported `func`: an `ODEFunc` (or any module exposing `gradient_net`, an nn.Sequential of stride-1
'same' Conv2d layers separated by ReLU, as built by `helpers.utils.create_convnet`).  The dynamics are
autonomous (`ODEFunc.forward` ignores t: modules/DiffEqSolver.py:77), so `t` only sets step sizes.

Semantics follow torchdiffeq 0.2.1: solution[0] = y0, float64 time, 'rk4' = 3/8 rule with one step per
output interval, strictly decreasing t integrates the negated dynamics on -t.
"""
import torch
import torch.nn as nn

from . import hip_ops

FIXED_GRID = ("euler", "midpoint", "rk4")
last_stats = hip_ops.LazyStats()  # nfe / n_accept / n_reject of the most recent dopri5 call (instrumentation; fills itself after an asynchronous solve)


def _host_times(t):
    return hip_ops.host_times(t)


def _check_monotone(t):
    if len(t) > 1:
        d = t[1:] - t[:-1]
        if not (bool((d > 0).all()) or bool((d < 0).all())):
            raise AssertionError("t must be strictly increasing or decreasing")


def conv_stack_of(func):
    """Find (and cache on the module) the packed conv stack behind an ODEFunc-like module."""
    cached = getattr(func, "_hip_stack", None)
    if cached is not None:
        return cached
    net = getattr(func, "gradient_net", None)
    if net is None and isinstance(func, nn.Sequential):
        net = func
    if not isinstance(net, nn.Sequential):
        raise TypeError("odeint(HIP): `func` must expose `gradient_net` (nn.Sequential of Conv2d/ReLU as built by "
                        "create_convnet); arbitrary Python dynamics are not supported and there is no CPU fallback")
    convs, final_tanh = [], False
    mods = list(net)
    for i, m in enumerate(mods):
        if isinstance(m, nn.Conv2d):
            convs.append(m)
        elif isinstance(m, nn.ReLU):
            continue
        elif isinstance(m, nn.Tanh) and i == len(mods) - 1:
            final_tanh = True
        else:
            raise TypeError(f"odeint(HIP): unsupported layer {m} in gradient_net (Conv2d/ReLU only)")
    # create_convnet alternates conv, act, conv, ... : every conv but the last is followed by ReLU
    for i, m in enumerate(mods[:-1]):
        if isinstance(m, nn.Conv2d) and not isinstance(mods[i + 1], nn.ReLU):
            raise TypeError("odeint(HIP): every hidden Conv2d must be followed by ReLU")
    stack = hip_ops.PackedConvStack(convs, final_tanh)
    try:
        object.__setattr__(func, "_hip_stack", stack)
    except Exception:
        pass
    return stack


def check_options(method, options, where="odeint"):
    """torchdiffeq's `options` that this path implements: none for the fixed-grid methods (one step per output interval; `step_size`,
    `grid_constructor`, `perturb` and `interp` would change the result and are refused rather than ignored), `first_step` and
    `max_num_steps` for dopri5."""
    known = set() if method in FIXED_GRID else {"first_step", "max_num_steps"}
    unknown = set(options or {}) - known
    if unknown:
        raise ValueError(f"{where}(HIP): unsupported {method} options {sorted(unknown)}")


def odeint(func, y0, t, rtol=1e-7, atol=1e-9, method=None, options=None):
    if method is None:
        method = "dopri5"
    if method not in ("euler", "midpoint", "rk4", "dopri5"):
        raise ValueError('Invalid method "{}". Must be one of euler, midpoint, rk4, dopri5'.format(method))
    check_options(method, options)
    hip_ops.require_device_tensor(y0, "y0")
    if torch.is_grad_enabled() and (y0.requires_grad or any(p.requires_grad for p in func.parameters())):
        from .autograd import odeint_with_grad
        return odeint_with_grad(func, y0, t, rtol, atol, method, options)
    return odeint_forward(func, y0, t, rtol, atol, method, options)


def odeint_forward(func, y0, t, rtol, atol, method, options=None):
    th = _host_times(t)
    _check_monotone(th)
    negate = False
    if len(th) > 1 and bool(th[0] > th[1]):  # torchdiffeq: strictly decreasing t => integrate -f on -t
        th = -th
        negate = True
    stack = conv_stack_of(func)
    if method in FIXED_GRID:
        check_options(method, options)
        return hip_ops.odeint_fixed(stack, method, y0, th, negate=negate)
    check_options(method, options)
    options = options or {}
    if hip_ops._async_dopri5 and not negate:   # enqueue only; the stats are read when somebody looks at them
        out, pending = hip_ops.odeint_dopri5_start(stack, y0, th, rtol, atol, first_step=float(options.get("first_step") or 0.0),
                                                   max_steps=int(options.get("max_num_steps") or 0))
        last_stats._bind(pending)
        return out
    out, stats = hip_ops.odeint_dopri5(stack, y0, th, rtol, atol, first_step=float(options.get("first_step") or 0.0),
                                       max_steps=int(options.get("max_num_steps") or 0), negate=negate)
    last_stats.clear()
    last_stats.update(stats)
    return out
