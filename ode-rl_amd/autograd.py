"""Autograd glue: `odeint` participates in autograd w.r.t. y0 and every parameter of the dynamics, as the reference's
does (ownership).  Fixed-grid methods: the exact gradient of the discrete solver (what the
reference's loss.backward() computes through torchdiffeq's ops), by the explicit reverse sweep of
csrc/fixed_grid.hip + csrc/wgrad.hip."""
import torch

from . import hip_ops


def _pinned(fn):
    """backward runs with the compute dtype its forward ran with (autocast is not active on the autograd thread)."""
    def wrapper(ctx, *grads):
        with hip_ops.compute_mode(ctx.mode):
            return fn(ctx, *grads)
    return wrapper


class _FixedGridOdeint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y0, t_host, method, stack, *params):
        ctx.mode = hip_ops.current_compute_dtype()
        out, ws = hip_ops.odeint_fixed(stack, method, y0.detach(), t_host, save=True)
        ctx.stack, ctx.method, ctx.t_host, ctx.batch, ctx.ws = stack, method, t_host, y0.shape[0], ws
        ctx.versions = tuple(p._version for p in params)
        ctx.params = params
        return out

    @staticmethod
    @_pinned
    def backward(ctx, grad_out):
        if tuple(p._version for p in ctx.params) != ctx.versions:
            raise RuntimeError("a parameter of the ODE dynamics was modified in place between forward and backward")
        gz0, gws, gbs = hip_ops.odeint_fixed_backward(ctx.stack, ctx.method, ctx.t_host, ctx.batch, grad_out, ctx.ws)
        ctx.ws = None
        grads = []
        for gw, gb in zip(gws, gbs):
            grads += [gw, gb]
        return (gz0, None, None, None) + tuple(grads)


class _Dopri5Odeint(torch.autograd.Function):
    """odeint(method="dopri5") under autograd: the gradient of the accepted steps (csrc/dopri5_backward.hip)."""

    @staticmethod
    def forward(ctx, y0, t_host, cfg, stack, *params):
        ctx.mode = hip_ops.current_compute_dtype()
        y0d = y0.detach()
        # the forward keeps the activations of the accepted steps when it can (64-channel fp32 stacks on the adaptive walk): the
        # backward is then the reverse sweep alone; otherwise ctx.saved is None and the backward re-integrates the logged steps
        out, stats, ctx.saved = hip_ops.odeint_dopri5_saving(stack, y0d, t_host, cfg["rtol"], cfg["atol"], first_step=cfg["first_step"],
                                                             max_steps=cfg["max_num_steps"])
        from .odeint import last_stats
        ctx.stack, ctx.t_host = stack, t_host
        ctx.versions = tuple(p._version for p in params)
        ctx.params = params
        if isinstance(stats, hip_ops.PendingDopri5):   # asynchronous solve: nothing is known yet -- the backward pass collects it
            last_stats._bind(stats)
            ctx.pending, ctx.accepted = stats, None
        else:
            ctx.pending = None
            last_stats.clear()
            last_stats.update(stats)
            if stats["n_accept"] > len(stats["accepted"]):
                raise RuntimeError(f"odeint(HIP, dopri5): {stats['n_accept']} accepted steps exceed the {hip_ops.LOG_CAP} "
                                   "the backward pass can re-integrate")
            ctx.accepted = stats["accepted"]
        ctx.save_for_backward(y0d)
        return out

    @staticmethod
    @_pinned
    def backward(ctx, grad_out):
        if tuple(p._version for p in ctx.params) != ctx.versions:
            raise RuntimeError("a parameter of the ODE dynamics was modified in place between forward and backward")
        (y0,) = ctx.saved_tensors
        if ctx.pending is not None:
            stats, ctx.saved = ctx.pending.collect()
            ctx.pending = None
            if stats["n_accept"] > len(stats["accepted"]):
                raise RuntimeError(f"odeint(HIP, dopri5): {stats['n_accept']} accepted steps exceed the {hip_ops.LOG_CAP} "
                                   "the backward pass can re-integrate")
            ctx.accepted = stats["accepted"]
        if ctx.saved is not None and len(ctx.accepted) >= 1:
            gz0, gws, gbs = hip_ops.odeint_dopri5_backward_saved(ctx.stack, ctx.t_host, ctx.accepted, grad_out, ctx.saved)
            ctx.saved = None
        else:
            gz0, gws, gbs = hip_ops.odeint_dopri5_backward(ctx.stack, ctx.t_host, ctx.accepted, y0, grad_out)
        grads = []
        for gw, gb in zip(gws, gbs):
            grads += [gw, gb]
        return (gz0, None, None, None) + tuple(grads)


class _CellFn(torch.autograd.Function):
    """One ConvGRU step under autograd (csrc/convgru_backward.hip: odehip_convgru_cell_backward)."""

    @staticmethod
    def forward(ctx, x, h, packed, *params):
        ctx.mode = hip_ops.current_compute_dtype()
        out = hip_ops.convgru_cell_forward(packed, x.detach(), h.detach())
        ctx.packed = packed
        ctx.versions = tuple(p._version for p in params)
        ctx.params = params
        ctx.save_for_backward(x.detach(), h.detach())
        return out

    @staticmethod
    @_pinned
    def backward(ctx, grad_out):
        if tuple(p._version for p in ctx.params) != ctx.versions:
            raise RuntimeError("a ConvGRUCell parameter was modified in place between forward and backward")
        x, h = ctx.saved_tensors
        gx, gh, grads = hip_ops.convgru_cell_backward(ctx.packed, x, h, grad_out)
        return (gx, gh, None) + tuple(grads)


def cell_with_grad(packed, x, h):
    return _CellFn.apply(x, h, packed, *packed._params())


class _EncodeFn(torch.autograd.Function):
    """ODEConvGRUCell.forward / run_ode_conv_gru under autograd: csrc/convgru_backward.hip keeps the per-frame conv outputs and
    sweeps back; the gradient may arrive through (mean, std) and, when latent_ys was asked for, through latent_ys."""

    @staticmethod
    def forward(ctx, inputs, timesteps, enc, want_latent, run_backwards, *params):
        ctx.mode = hip_ops.current_compute_dtype()
        mean, std, latent, saved = hip_ops.odeconvgru_encode_train(enc, inputs.detach(), timesteps, want_latent, run_backwards)
        ctx.enc, ctx.saved, ctx.want_latent = enc, saved, want_latent
        ctx.versions = tuple(p._version for p in params)
        ctx.params = params
        return (mean, std, latent) if want_latent else (mean, std)

    @staticmethod
    @_pinned
    def backward(ctx, grad_mean, grad_std, grad_latent=None):
        if tuple(p._version for p in ctx.params) != ctx.versions:
            raise RuntimeError("an encoder parameter was modified in place between forward and backward")
        if ctx.saved is None:
            raise RuntimeError("the encoder's saved activations were already consumed (backward called twice)")
        gin, grads = hip_ops.odeconvgru_encode_backward(ctx.enc, ctx.saved, grad_mean, grad_std, grad_latent if ctx.want_latent else None)
        ctx.saved = None
        return (gin, None, None, None, None) + tuple(grads)


def encode_with_grad(enc, inputs, timesteps, want_latent=False, run_backwards=True):
    return _EncodeFn.apply(inputs, timesteps, enc, bool(want_latent), bool(run_backwards), *hip_ops.encoder_params(enc))


class _WarpCompositeFn(torch.autograd.Function):
    """VidODE's warp chain + mask compositing (csrc/warp.hip): one launch forward, one backward."""

    @staticmethod
    def forward(ctx, pred_outputs, start_image, grid_x, grid_y):
        po, st = pred_outputs.detach().contiguous(), start_image.detach().contiguous()
        gx, gy = grid_x.detach().contiguous(), grid_y.detach().contiguous()
        pred_x, warped, masks = hip_ops.warp_composite(po, st, gx, gy)
        ctx.save_for_backward(po, st, warped, gx, gy)
        return pred_x, warped, masks

    @staticmethod
    def backward(ctx, g_pred_x, g_warped, g_masks):
        po, st, warped, gx, gy = ctx.saved_tensors
        g_po, g_start = hip_ops.warp_composite_backward(po, st, warped, gx, gy, g_pred_x, g_warped, g_masks, ctx.needs_input_grad[1])
        return g_po, g_start, None, None


def warp_composite(pred_outputs, start_image, grid_x, grid_y):
    """(pred_x, warped_pred_x, pred_masks) of models/VidODE.py:119-138 from the flow decoder's output and the last observed frame."""
    if torch.is_grad_enabled() and (pred_outputs.requires_grad or start_image.requires_grad):
        return _WarpCompositeFn.apply(pred_outputs, start_image, grid_x, grid_y)
    return hip_ops.warp_composite(pred_outputs, start_image, grid_x, grid_y)


class _Upsample2xFn(torch.autograd.Function):
    """Bilinear x2 upsampling (csrc/upsample.hip): linear, so the backward needs nothing saved."""

    @staticmethod
    def forward(ctx, x):
        return hip_ops.upsample2x(x.detach())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        return hip_ops.upsample2x_backward(g)


def upsample2x(x):
    if torch.is_grad_enabled() and x.requires_grad:
        return _Upsample2xFn.apply(x)
    return hip_ops.upsample2x(x)


class _BnReluUpFn(torch.autograd.Function):
    """BatchNorm2d -> ReLU (-> bilinear x2 upsampling) in one pass (csrc/bn_relu_up.hip): VidODE's flow decoder, models/VidODE.py:34-36."""

    @staticmethod
    def forward(ctx, x, weight, bias, conv_bias, bn, upsample):
        out, saved = hip_ops.bn_relu_up_forward(x.detach(), bn, upsample, conv_bias=conv_bias)
        ctx.saved_stats, ctx.upsample = saved, upsample
        ctx.affine, ctx.has_conv_bias = weight is not None, conv_bias is not None
        ctx.save_for_backward(x.detach())
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        gx, gw, gb = hip_ops.bn_relu_up_backward(g, x, ctx.saved_stats, ctx.upsample)
        gcb = None
        if ctx.has_conv_bias:
            # a constant in front of batch statistics has no gradient; in eval() mode d/db = sum of dx = scale * sum g_pre = scale * d beta
            stats, training = ctx.saved_stats
            gcb = torch.zeros_like(gb) if training else stats[2] * gb
        return gx, (gw if ctx.affine else None), (gb if ctx.affine else None), gcb, None, None


def bn_relu_up(x, bn, upsample, conv_bias=None):
    """relu(bn(x)), upsampled x2 if `upsample`; bn: nn.BatchNorm2d (train() or eval() as the module says; running statistics and
    num_batches_tracked updated as the module itself would).  conv_bias: the bias of the convolution that produced x if the caller
    left it out of the convolution (hip_ops.bn_relu_up_forward)."""
    if torch.is_grad_enabled() and (x.requires_grad or (bn.weight is not None and bn.weight.requires_grad) or
                                    (conv_bias is not None and conv_bias.requires_grad)):
        return _BnReluUpFn.apply(x, bn.weight, bn.bias, conv_bias, bn, bool(upsample))
    return hip_ops.bn_relu_up_forward(x, bn, upsample, conv_bias=conv_bias)[0]


class _AdjointOdeint(torch.autograd.Function):
    """torchdiffeq.odeint_adjoint: forward without a graph, backward by integrating the adjoint ODE backwards."""

    @staticmethod
    def forward(ctx, y0, t_host, method, stack, *params):
        ctx.mode = hip_ops.current_compute_dtype()
        out = hip_ops.odeint_fixed(stack, method, y0.detach(), t_host)
        ctx.stack, ctx.method, ctx.t_host = stack, method, t_host
        ctx.versions = tuple(p._version for p in params)
        ctx.params = params
        ctx.save_for_backward(out)
        return out

    @staticmethod
    @_pinned
    def backward(ctx, grad_out):
        if tuple(p._version for p in ctx.params) != ctx.versions:
            raise RuntimeError("a parameter of the ODE dynamics was modified in place between forward and backward")
        (y_traj,) = ctx.saved_tensors
        gz0, gws, gbs = hip_ops.odeint_adjoint_backward(ctx.stack, ctx.method, ctx.t_host, y_traj, grad_out)
        grads = []
        for gw, gb in zip(gws, gbs):
            grads += [gw, gb]
        return (gz0, None, None, None) + tuple(grads)


class _AdjointDopri5(torch.autograd.Function):
    """odeint_adjoint with method="dopri5": adaptive forward (csrc/dopri5.hip), adaptive augmented backward
    (csrc/adjoint_dopri5.hip) under the seminorm."""

    @staticmethod
    def forward(ctx, y0, t_host, cfg, stack, *params):
        ctx.mode = hip_ops.current_compute_dtype()
        from .odeint import last_stats
        if hip_ops._async_dopri5:
            out, pending = hip_ops.odeint_dopri5_start(stack, y0.detach(), t_host, cfg["rtol"], cfg["atol"], first_step=cfg["first_step"],
                                                       max_steps=cfg["max_num_steps"])
            last_stats._bind(pending)
            ctx.pending = pending
        else:
            out, stats = hip_ops.odeint_dopri5(stack, y0.detach(), t_host, cfg["rtol"], cfg["atol"],
                                               first_step=cfg["first_step"], max_steps=cfg["max_num_steps"])
            last_stats.clear()
            last_stats.update(stats)
            ctx.pending = None
        ctx.stack, ctx.t_host, ctx.cfg = stack, t_host, cfg
        ctx.versions = tuple(p._version for p in params)
        ctx.params = params
        ctx.save_for_backward(out)
        return out

    @staticmethod
    @_pinned
    def backward(ctx, grad_out):
        if tuple(p._version for p in ctx.params) != ctx.versions:
            raise RuntimeError("a parameter of the ODE dynamics was modified in place between forward and backward")
        (y_traj,) = ctx.saved_tensors
        if ctx.pending is not None:   # the forward solve's outcome: an error (or a sealed, unfinished solve) must stop the adjoint here
            ctx.pending.collect()
            ctx.pending = None
        cfg = ctx.cfg
        stats = {}
        gz0, gws, gbs = hip_ops.odeint_adjoint_dopri5_backward(ctx.stack, ctx.t_host, y_traj, grad_out, cfg["adjoint_rtol"],
                                                               cfg["adjoint_atol"], max_accept=cfg["max_accept"], stats=stats,
                                                               mixed_norm=cfg["mixed_norm"])
        last_adjoint_stats.clear()
        last_adjoint_stats.update(stats)
        grads = []
        for gw, gb in zip(gws, gbs):
            grads += [gw, gb]
        return (gz0, None, None, None) + tuple(grads)


last_adjoint_stats = {}


def odeint_adjoint(func, y0, t, rtol=1e-7, atol=1e-9, method=None, options=None, adjoint_rtol=None, adjoint_atol=None,
                   adjoint_method=None, adjoint_options=None, adjoint_params=None):
    """`torchdiffeq.odeint_adjoint(func, y0, t, rtol=, atol=, method=, adjoint_options=)`.

    Fixed-grid methods: one backward step of the same method per interval.  dopri5: adaptive backward solve with
    torchdiffeq's default mixed norm (every parameter tensor's error ratio steers the steps too) or, with
    `adjoint_options={"norm": "seminorm"}`, with the cheaper seminorm; `max_accept` in adjoint_options bounds the accepted
    backward steps whose activations are kept."""
    from .odeint import FIXED_GRID, _check_monotone, _host_times, check_options, conv_stack_of, odeint
    if method is None:
        method = "dopri5"
    if method not in FIXED_GRID and method != "dopri5":
        raise ValueError('Invalid method "{}". Must be one of euler, midpoint, rk4, dopri5'.format(method))
    if adjoint_method is not None and adjoint_method != method:
        raise NotImplementedError("odeint_adjoint(HIP): adjoint_method must equal method")
    check_options(method, options, "odeint_adjoint")
    if adjoint_params is not None and ({id(p) for p in adjoint_params} != {id(p) for p in func.parameters()}):
        # torchdiffeq integrates a_theta for exactly these tensors; this path always does so for every parameter of `func`
        raise NotImplementedError("odeint_adjoint(HIP): adjoint_params must be omitted or be all of func.parameters()")
    if method in FIXED_GRID and adjoint_options:
        raise ValueError(f"odeint_adjoint(HIP): unsupported {method} adjoint_options {sorted(adjoint_options)}")
    if not torch.is_grad_enabled() or not (y0.requires_grad or any(p.requires_grad for p in func.parameters())):
        return odeint(func, y0, t, rtol=rtol, atol=atol, method=method, options=options)
    hip_ops.require_device_tensor(y0, "y0")
    th = _host_times(t)
    _check_monotone(th)
    if len(th) > 1 and bool(th[0] > th[1]):
        raise NotImplementedError("odeint_adjoint(HIP): decreasing time grids are not supported")
    stack = conv_stack_of(func)
    params = []
    for c in stack.convs:
        params += [c.weight, c.bias]
    if method in FIXED_GRID:
        return _AdjointOdeint.apply(y0, th, method, stack, *params)
    adjoint_options = dict(adjoint_options or {})
    norm = adjoint_options.get("norm")
    if norm not in (None, "mixed", "seminorm"):
        raise ValueError(f'odeint_adjoint(HIP): adjoint_options["norm"] must be "seminorm" or omitted (torchdiffeq\'s mixed norm); got {norm!r}')
    unknown = set(adjoint_options) - {"norm", "max_accept"}
    if unknown:
        raise ValueError(f"odeint_adjoint(HIP): unsupported adjoint_options {sorted(unknown)}")
    options = options or {}
    cfg = dict(rtol=float(rtol), atol=float(atol), first_step=float(options.get("first_step") or 0.0),
               max_num_steps=int(options.get("max_num_steps") or 0),
               adjoint_rtol=float(rtol if adjoint_rtol is None else adjoint_rtol),
               adjoint_atol=float(atol if adjoint_atol is None else adjoint_atol), max_accept=adjoint_options.get("max_accept"),
               mixed_norm=norm != "seminorm")
    return _AdjointDopri5.apply(y0, th, cfg, stack, *params)


def odeint_with_grad(func, y0, t, rtol, atol, method, options=None):
    from .odeint import FIXED_GRID, _check_monotone, _host_times, check_options, conv_stack_of
    check_options(method, options)
    th = _host_times(t)
    _check_monotone(th)
    if len(th) > 1 and bool(th[0] > th[1]):
        raise NotImplementedError("odeint(HIP): reversed-time integration is not implemented yet")
    stack = conv_stack_of(func)
    params = []
    for c in stack.convs:
        params += [c.weight, c.bias]
    if method in FIXED_GRID:
        return _FixedGridOdeint.apply(y0, th, method, stack, *params)
    options = options or {}
    cfg = dict(rtol=float(rtol), atol=float(atol), first_step=float(options.get("first_step") or 0.0),
               max_num_steps=int(options.get("max_num_steps") or 0))
    return _Dopri5Odeint.apply(y0, th, cfg, stack, *params)
