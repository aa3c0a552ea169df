"""Autograd glue: `odeint` participates in autograd w.r.t. y0 and every parameter of the dynamics, as the reference's
does (ownership).  Fixed-grid methods: the exact gradient of the discrete solver (what the
reference's loss.backward() computes through torchdiffeq's ops), by the explicit reverse sweep of
csrc/fixed_grid.hip + csrc/wgrad.hip."""
import torch

from . import hip_ops


class _FixedGridOdeint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y0, t_host, method, stack, *params):
        out, ws = hip_ops.odeint_fixed(stack, method, y0.detach(), t_host, save=True)
        ctx.stack, ctx.method, ctx.t_host, ctx.batch, ctx.ws = stack, method, t_host, y0.shape[0], ws
        ctx.versions = tuple(p._version for p in params)
        ctx.params = params
        return out

    @staticmethod
    def backward(ctx, grad_out):
        if tuple(p._version for p in ctx.params) != ctx.versions:
            raise RuntimeError("a parameter of the ODE dynamics was modified in place between forward and backward")
        gz0, gws, gbs = hip_ops.odeint_fixed_backward(ctx.stack, ctx.method, ctx.t_host, ctx.batch, grad_out, ctx.ws)
        ctx.ws = None
        grads = []
        for gw, gb in zip(gws, gbs):
            grads += [gw, gb]
        return (gz0, None, None, None) + tuple(grads)


class _AdjointOdeint(torch.autograd.Function):
    """torchdiffeq.odeint_adjoint: forward without a graph, backward by integrating the adjoint ODE backwards."""

    @staticmethod
    def forward(ctx, y0, t_host, method, stack, *params):
        out = hip_ops.odeint_fixed(stack, method, y0.detach(), t_host)
        ctx.stack, ctx.method, ctx.t_host = stack, method, t_host
        ctx.versions = tuple(p._version for p in params)
        ctx.params = params
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        if tuple(p._version for p in ctx.params) != ctx.versions:
            raise RuntimeError("a parameter of the ODE dynamics was modified in place between forward and backward")
        (y_traj,) = ctx.saved_tensors
        gz0, gws, gbs = hip_ops.odeint_adjoint_backward(ctx.stack, ctx.method, ctx.t_host, y_traj, grad_out)
        grads = []
        for gw, gb in zip(gws, gbs):
            grads += [gw, gb]
        return (gz0, None, None, None) + tuple(grads)


def odeint_adjoint(func, y0, t, rtol=1e-7, atol=1e-9, method=None, options=None, adjoint_params=None):
    """`torchdiffeq.odeint_adjoint(func, y0, t, rtol=, atol=, method=)` for the fixed-grid methods."""
    from .odeint import FIXED_GRID, _check_monotone, _host_times, conv_stack_of, odeint
    if method is None:
        method = "dopri5"
    if not torch.is_grad_enabled() or not (y0.requires_grad or any(p.requires_grad for p in func.parameters())):
        return odeint(func, y0, t, rtol=rtol, atol=atol, method=method, options=options)
    if method not in FIXED_GRID:
        raise NotImplementedError("odeint_adjoint(HIP): the adaptive (dopri5) adjoint is not implemented yet; "
                                  "fixed-grid methods (euler, midpoint, rk4) are")
    hip_ops.require_device_tensor(y0, "y0")
    th = _host_times(t)
    _check_monotone(th)
    if len(th) > 1 and bool(th[0] > th[1]):
        raise NotImplementedError("odeint_adjoint(HIP): decreasing time grids are not supported")
    stack = conv_stack_of(func)
    params = []
    for c in stack.convs:
        params += [c.weight, c.bias]
    return _AdjointOdeint.apply(y0, th, method, stack, *params)


def odeint_with_grad(func, y0, t, rtol, atol, method, options=None):
    from .odeint import FIXED_GRID, _check_monotone, _host_times, conv_stack_of
    th = _host_times(t)
    _check_monotone(th)
    if len(th) > 1 and bool(th[0] > th[1]):
        raise NotImplementedError("odeint(HIP): reversed-time integration is not implemented yet")
    if method not in FIXED_GRID:
        raise NotImplementedError("odeint(HIP): backward through dopri5 is not implemented yet (use a fixed-grid method)")
    stack = conv_stack_of(func)
    params = []
    for c in stack.convs:
        params += [c.weight, c.bias]
    return _FixedGridOdeint.apply(y0, th, method, stack, *params)
