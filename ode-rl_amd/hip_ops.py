"""Thin torch-facing wrappers over the C ABI: pointer plumbing, workspace and packed-weight caches.

Nothing here computes: every function hands raw device pointers to libodecgru_hip.so on torch's
current stream.  Tensors must be CUDA fp32; anything else raises (there is no CPU fallback).
"""
import ctypes
import os
import threading

import torch

from . import _lib

_workspaces = {}


_last_stream = None   # the stream the library's process-global state (workspace caches, flag areas, layer tables) was last used on


def _stream():
    """The caller's current stream, handed to the C ABI.  The library's state is process-global (ONE
    process per GPU, one stream at a time): calls on a second DEVICE are refused, and when the current stream CHANGES the new
    stream first waits for everything enqueued on the previous one, so two streams can never run library launches that share a
    workspace or flag area concurrently.  (Under graph capture the wait cannot be expressed -- an event from outside the capture
    -- and the capturing caller has ordered its streams itself.)"""
    global _last_stream
    cur = torch.cuda.current_stream()
    if _last_stream is None:
        _last_stream = cur
    elif cur != _last_stream:
        if cur.device != _last_stream.device:
            raise RuntimeError(f"ode_rl_amd drives ONE GPU per process: first used on {_last_stream.device}, now called on {cur.device}")
        if not torch.cuda.is_current_stream_capturing():
            cur.wait_stream(_last_stream)
        _last_stream = cur
    return ctypes.c_void_p(cur.cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def require_device_tensor(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} is on {t.device}: the HIP path needs CUDA (ROCm) tensors and has no CPU fallback")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32 (got {t.dtype})")


_CANARY = 4096          # guard bytes behind every workspace: a kernel that overruns its carve-out trips check_canaries()
_guarded = []           # weak references to (full buffer, payload bytes)


def alloc_workspace(nbytes, device):
    """A fresh workspace of `nbytes` with a guard region behind it; returns the payload view (numel() == nbytes)."""
    import weakref
    nbytes = max(int(nbytes), 1024)
    full = torch.empty(nbytes + _CANARY, dtype=torch.uint8, device=device)
    full[nbytes:].fill_(0xA5)
    view = full[:nbytes]
    if len(_guarded) >= 64:   # forget workspaces that have been freed (one is allocated per training-mode forward)
        _guarded[:] = [(r, n) for r, n in _guarded if r() is not None]
    _guarded.append((weakref.ref(full), nbytes))
    view._odehip_full = full   # keeps the guard alive as long as the view
    return view


def check_canaries():
    """Raises if any live workspace's guard bytes were overwritten (tests call this after every GPU test)."""
    alive = []
    for ref, nbytes in _guarded:
        full = ref()
        if full is None:
            continue
        alive.append((ref, nbytes))
        if not bool((full[nbytes:] == 0xA5).all()):
            raise RuntimeError(f"a HIP kernel wrote beyond its {nbytes}-byte workspace")
    _guarded[:] = alive


def workspace(key, nbytes, device):
    """Persistent per-shape scratch so that pointers stay stable across calls (graph replay)."""
    k = (key, device)
    buf = _workspaces.get(k)
    if buf is None or buf.numel() < nbytes:
        buf = alloc_workspace(nbytes, device)
        _workspaces[k] = buf
    return buf


_t_cache = {}


def host_times(t):
    """float64 host copy of a 1-D time tensor.  A device-resident `t` costs a device->host copy (a stream synchronisation);
    it is cached per (base tensor OBJECT, version, view geometry): the base is held by weak reference, so a new tensor that
    happens to reuse the address of a freed one can never hit the cache."""
    import weakref
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(t, dtype=torch.float64)
    if t.dim() != 1:
        raise AssertionError("`t` must be one dimensional")
    if not t.is_cuda:
        return t.detach().to(torch.float64)
    base = t._base if t._base is not None else t
    key = (id(base), base._version, t.storage_offset(), t.numel(), t.stride(0) if t.numel() > 1 else 1, t.dtype)
    hit = _t_cache.get(key)
    if hit is not None and hit[0]() is base:
        return hit[1]
    if len(_t_cache) > 64:
        _t_cache.clear()
    host = t.detach().to("cpu", torch.float64)
    _t_cache[key] = (weakref.ref(base), host)
    return host


def nchw_to_q4(x):
    require_device_tensor(x, "x")
    x = x.contiguous()
    b, c, h, w = x.shape
    if (h, w) != (16, 16):
        raise ValueError(f"latent maps must be 16x16 (got {h}x{w})")
    out = torch.empty((b, c // 4, 256, 4), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().odehip_nchw_to_q4(_ptr(x), _ptr(out), b, c, _stream()))
    return out


def q4_to_nchw(x):
    require_device_tensor(x, "x")
    b, q = x.shape[0], x.shape[1]
    out = torch.empty((b, q * 4, 16, 16), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().odehip_q4_to_nchw(_ptr(x), _ptr(out), b, q * 4, _stream()))
    return out


def pack_conv_weight(w, transpose_flip=False):
    """(Cout,Cin,k,k) fp32 -> MFMA-ordered image.  With transpose_flip the result is the packed weight of
    the input-gradient conv (Cin and Cout swap roles)."""
    require_device_tensor(w, "weight")
    w = w.detach().contiguous()
    co, ci, k, k2 = w.shape
    if k != k2:
        raise ValueError("square kernels only")
    if transpose_flip:
        co, ci = ci, co
    out = torch.empty(co * ci * k * k, dtype=torch.float32, device=w.device)
    _lib.check(_lib.load().odehip_pack_conv_weight(_ptr(w), _ptr(out), co, ci, k, int(bool(transpose_flip)), _stream()))
    return out


def pack_conv_weight_winograd(w, transpose_flip=False):
    """(Cout,Cin,3,3) fp32 -> Winograd F(2x2,3x3) image U = G g G^T (16 values per channel pair)."""
    require_device_tensor(w, "weight")
    w = w.detach().contiguous()
    co, ci, k, k2 = w.shape
    if (k, k2) != (3, 3):
        raise ValueError("Winograd form exists for 3x3 kernels only")
    if transpose_flip:
        co, ci = ci, co
    out = torch.empty(co * ci * 16, dtype=torch.float32, device=w.device)
    _lib.check(_lib.load().odehip_pack_conv_weight_winograd(_ptr(w), _ptr(out), co, ci, int(bool(transpose_flip)), _stream()))
    return out


def pack_conv_weights_many(jobs):
    """jobs: [(weight (cout,cin,k,k), kind, transpose_flip: bool)], kind 0 / False = pack_conv_weight, 1 / True =
    pack_conv_weight_winograd (3x3), 2 = pack_conv_weight_winograd5 (5x5) -> list of packed tensors, identical to those calls' results
    but ONE launch per 32 jobs (odehip_pack_conv_weights)."""
    outs, recs = [], []
    for w, kind, tf in jobs:
        kind = int(kind)
        require_device_tensor(w, "conv weight")
        w = w.detach().contiguous()
        if w.dtype != torch.float32:
            raise ValueError("conv weights must be float32")
        co, ci, k, _ = w.shape
        if tf:
            co, ci = ci, co
        out = torch.empty(co * ci * (16 if kind == 1 else (36 if kind == 2 else k * k)), dtype=torch.float32, device=w.device)
        outs.append(out)
        recs.append((w, out, co, ci, k, kind, 1 if tf else 0))
    lib = _lib.load()
    for i in range(0, len(recs), _lib.MAX_PACK_JOBS):
        chunk = recs[i:i + _lib.MAX_PACK_JOBS]
        arr = (_lib.PackJob * len(chunk))()
        for j, (w, out, co, ci, k, kind, tf) in enumerate(chunk):
            arr[j].w, arr[j].out, arr[j].cout, arr[j].cin, arr[j].ks, arr[j].kind, arr[j].transpose_flip = _ptr(w), _ptr(out), co, ci, k, kind, tf
        _lib.check(lib.odehip_pack_conv_weights(ctypes.cast(arr, ctypes.c_void_p), len(chunk), _stream()))
    return outs


def pack_conv_weight_winograd5(w, transpose_flip=False):
    """(Cout,Cin,5,5) fp32 -> Winograd F(2x2,5x5) image U = G g G^T (36 values per channel pair; csrc/conv_wino5.hip)."""
    require_device_tensor(w, "weight")
    w = w.detach().contiguous()
    co, ci, k, k2 = w.shape
    if (k, k2) != (5, 5):
        raise ValueError("this Winograd form is for 5x5 kernels")
    if transpose_flip:
        co, ci = ci, co
    out = torch.empty(co * ci * 36, dtype=torch.float32, device=w.device)
    _lib.check(_lib.load().odehip_pack_conv_weight_winograd5(_ptr(w), _ptr(out), co, ci, int(bool(transpose_flip)), _stream()))
    return out


def winograd5_enabled():
    """F(2x2,5x5) for the ConvGRU's fp32 5x5 convolutions: on by default, ODEHIP_WINO5=0 runs the direct kernel (A/B, tests)."""
    return os.environ.get("ODEHIP_WINO5", "1") != "0"


# ---- compute dtype of the 3x3 conv layers: "f32" (exact fp32 MFMA, default) or "bf16" (bf16 operands, fp32 accumulation;
# state, stage combines and conv outputs stay fp32).  bf16 is chosen by set_compute_dtype("bf16") or, as the reference's
# users would ask for it, by running under torch.autocast(device_type="cuda", dtype=torch.bfloat16).
_global_mode = None
_forced = threading.local()


def set_compute_dtype(mode):
    """"f32", "bf16", or None (= follow torch.autocast)."""
    global _global_mode
    if mode not in (None, "f32", "bf16"):
        raise ValueError('compute dtype must be "f32", "bf16" or None')
    _global_mode = mode


def current_compute_dtype():
    forced = getattr(_forced, "mode", None)
    if forced is not None:
        return forced
    if _global_mode is not None:
        return _global_mode
    try:
        on, dt = torch.is_autocast_enabled("cuda"), torch.get_autocast_dtype("cuda")
    except TypeError:   # older signatures
        on, dt = torch.is_autocast_enabled(), torch.get_autocast_gpu_dtype()
    return "bf16" if (on and dt == torch.bfloat16) else "f32"


class compute_mode:
    """Pins the compute dtype inside a backward pass to the one its forward ran with (autocast is not active on the autograd
    engine's thread)."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = getattr(_forced, "mode", None)
        _forced.mode = self.mode

    def __exit__(self, *exc):
        _forced.mode = self.prev


def pack_conv_weight_bf16(w, transpose_flip=False):
    """(Cout,Cin,3,3) fp32 -> bf16 A-operand image of the bf16 MFMA kernel (round to nearest even)."""
    require_device_tensor(w, "weight")
    w = w.detach().contiguous()
    co, ci, k, k2 = w.shape
    if (k, k2) != (3, 3):
        raise ValueError("the bf16 kernel serves 3x3 layers only")
    if transpose_flip:
        co, ci = ci, co
    out = torch.empty(co * ci * 9, dtype=torch.bfloat16, device=w.device)
    _lib.check(_lib.load().odehip_pack_conv_weight_bf16(_ptr(w), _ptr(out), co, ci, int(bool(transpose_flip)), _stream()))
    return out


def pack_conv_weight_bf16_ks(w, transpose_flip=False):
    """(Cout,Cin,5,5) fp32 -> block-major bf16 image of the 5x5 bf16 ring kernel."""
    require_device_tensor(w, "weight")
    w = w.detach().contiguous()
    co, ci, k, k2 = w.shape
    if transpose_flip:
        co, ci = ci, co
    out = torch.empty(co * ci * k * k2, dtype=torch.bfloat16, device=w.device)
    _lib.check(_lib.load().odehip_pack_conv_weight_bf16_ks(_ptr(w), _ptr(out), co, ci, k, int(bool(transpose_flip)), _stream()))
    return out


USE_FUSED = os.environ.get("ODEHIP_NO_FUSED") is None   # bf16 mode, 64-channel 3x3 stacks: whole f in one launch (fstack_bf16.hip)


def pack_fused_bf16(convs, reverse_transposed=False):
    """Fused bf16 image of a stack of 64 -> 64 3x3 convs in execution order (the input-gradient chain runs the layers
    backwards with transposed + flipped weights)."""
    lib = _lib.load()
    n = len(convs)
    out = torch.empty(lib.odehip_fused_bf16_weight_bytes(n), dtype=torch.uint8, device=convs[0].weight.device)
    for e in range(n):
        c = convs[n - 1 - e] if reverse_transposed else convs[e]
        w = c.weight.detach().contiguous()
        require_device_tensor(w, "conv weight")
        _lib.check(lib.odehip_pack_convstack_fused_bf16(_ptr(w), _ptr(out), e, int(bool(reverse_transposed)), _stream()))
    return out


def _fusable(convs, ks):
    return USE_FUSED and ks == 3 and all(c.in_channels == 64 and c.out_channels == 64 for c in convs)


def _bf16_cell_ok(cell_input, hidden, ks):
    return ks == 5 and cell_input % 16 == 0 and hidden % 32 == 0 and cell_input + hidden <= 128


def _bf16_ok(cin, cout, ks):
    return ks == 3 and cin % 16 == 0 and cin <= 128 and cin // 16 in (1, 2, 4, 8) and cout % 32 == 0


USE_WINOGRAD = os.environ.get("ODEHIP_NO_WINOGRAD") is None   # 3x3 layers with cin % 16 == 0 run the Winograd kernel (2.25x fewer MFMAs, still exact-fp32 arithmetic)


def conv_q4(src1, w_packed, bias, cout, ks, src2=None, relu=False, w_wino=None, w_bf16=None):
    """One conv layer on Q4 activations (tests / building block)."""
    require_device_tensor(src1, "src1")
    b = src1.shape[0]
    cin1 = src1.shape[1] * 4
    cin = cin1 + (src2.shape[1] * 4 if src2 is not None else 0)
    dst = torch.empty((b, cout // 4, 256, 4), dtype=torch.float32, device=src1.device)
    d = _lib.ConvDesc(src1=src1.data_ptr(), src2=src2.data_ptr() if src2 is not None else None, cin1=cin1, cin=cin,
                      cout=cout, ks=ks, batch=b, w_packed=w_packed.data_ptr(),
                      w_wino=w_wino.data_ptr() if w_wino is not None else None,
                      w_bf16=w_bf16.data_ptr() if w_bf16 is not None else None,
                      bias=bias.data_ptr() if bias is not None else None,
                      dst=dst.data_ptr(), relu=int(relu))
    _lib.check(_lib.load().odehip_conv_q4(ctypes.byref(d), _stream()))
    return dst


class PackedConvStack:
    """Packed weights of a `create_convnet` Sequential, refreshed when the live Parameters change
    (the optimizer updates them in place between calls: SURVEY.md section 8b, ownership)."""

    def __init__(self, convs, final_tanh=False):
        self.convs = list(convs)
        self.final_tanh = bool(final_tanh)
        self._cache = {}      # compute dtype -> dict(stamp, desc, keep, dgrad)
        self._bias = None
        self.desc = None

    def _current_stamp(self):
        return tuple((c.weight.data_ptr(), c.weight._version, c.bias.data_ptr(), c.bias._version) for c in self.convs)

    def refresh(self, mode=None):
        mode = mode or current_compute_dtype()
        stamp = self._current_stamp()
        ent = self._cache.get(mode)
        if ent is not None and ent["stamp"] == stamp:
            self.desc = ent["desc"]
            self._bias = ent["bias"]
            return ent["desc"]
        convs = self.convs
        if len(convs) > _lib.MAX_LAYERS:
            raise ValueError(f"at most {_lib.MAX_LAYERS} conv layers are supported")
        ks = convs[0].kernel_size[0]
        for c in convs:
            if c.kernel_size != (ks, ks) or c.stride != (1, 1) or c.padding != (ks // 2, ks // 2) or c.dilation != (1, 1) \
                    or c.groups != 1 or c.bias is None:
                raise ValueError("the HIP path supports stride-1 'same' square convs with bias only "
                                 f"(got {c}); downsize=True dynamics are not supported")
            require_device_tensor(c.weight, "conv weight")
        wino_ok = USE_WINOGRAD and ks == 3
        want_wino = [wino_ok and c.in_channels % 16 == 0 for c in convs]
        both = pack_conv_weights_many([(c.weight, False, False) for c in convs] + [(c.weight, True, False) for c, ww in zip(convs, want_wino) if ww])
        packed, rest = both[:len(convs)], iter(both[len(convs):])
        wino = [next(rest) if ww else None for ww in want_wino]
        bf16 = [pack_conv_weight_bf16(c.weight) if mode == "bf16" and _bf16_ok(c.in_channels, c.out_channels, ks) else None
                for c in convs]
        bias = [c.bias.detach().contiguous() for c in convs]
        fused = pack_fused_bf16(convs) if mode == "bf16" and _fusable(convs, ks) else None
        d = _lib.ConvStack()
        d.w_fused = fused.data_ptr() if fused is not None else None
        d.n_convs = len(convs)
        d.ks = ks
        d.channels[0] = convs[0].in_channels
        for i, c in enumerate(convs):
            d.channels[i + 1] = c.out_channels
            d.w_packed[i] = packed[i].data_ptr()
            d.w_wino[i] = wino[i].data_ptr() if wino[i] is not None else None
            d.w_bf16[i] = bf16[i].data_ptr() if bf16[i] is not None else None
            d.bias[i] = bias[i].data_ptr()
        d.final_tanh = int(self.final_tanh)
        self._cache[mode] = dict(stamp=stamp, desc=d, keep=(packed, wino, bf16, fused), bias=bias, dgrad=None)
        self.desc, self._bias = d, bias
        return d

    def dgrad_desc(self, mode=None):
        """Stack of the input-gradient convs (weights packed transposed + flipped), built on first use."""
        mode = mode or current_compute_dtype()
        d0 = self.refresh(mode)
        ent = self._cache[mode]
        if ent["dgrad"] is None:
            want_wino = [USE_WINOGRAD and d0.ks == 3 and c.out_channels % 16 == 0 for c in self.convs]
            both = pack_conv_weights_many([(c.weight, False, True) for c in self.convs] +
                                          [(c.weight, True, True) for c, ww in zip(self.convs, want_wino) if ww])
            packed, rest = both[:len(self.convs)], iter(both[len(self.convs):])
            wino = [next(rest) if ww else None for ww in want_wino]
            bf16 = [pack_conv_weight_bf16(c.weight, transpose_flip=True)
                    if mode == "bf16" and _bf16_ok(c.out_channels, c.in_channels, d0.ks) else None for c in self.convs]
            fused = pack_fused_bf16(self.convs, reverse_transposed=True) if mode == "bf16" and _fusable(self.convs, d0.ks) else None
            d = _lib.ConvStack()
            d.w_fused = fused.data_ptr() if fused is not None else None
            d.n_convs, d.ks = d0.n_convs, d0.ks
            for i in range(len(self.convs) + 1):
                d.channels[i] = d0.channels[i]
            for i, p in enumerate(packed):
                d.w_packed[i] = p.data_ptr()
                d.w_wino[i] = wino[i].data_ptr() if wino[i] is not None else None
                d.w_bf16[i] = bf16[i].data_ptr() if bf16[i] is not None else None
                d.bias[i] = ent["bias"][i].data_ptr()
            ent["dgrad"] = (d, packed, wino, bf16, fused)
        return ent["dgrad"][0]


def convstack_forward(stack, y, negate=False):
    require_device_tensor(y, "y")
    desc = stack.refresh()
    y = y.contiguous()
    b = y.shape[0]
    if y.dim() != 4 or tuple(y.shape[2:]) != (16, 16) or y.shape[1] != desc.channels[0]:
        raise ValueError(f"y must be (B,{desc.channels[0]},16,16) (got {tuple(y.shape)})")
    lib = _lib.load()
    nbytes = lib.odehip_convstack_workspace_bytes(ctypes.byref(desc), b)
    ws = workspace(("f", b, tuple(desc.channels)), nbytes, y.device)
    out = torch.empty((b, desc.channels[desc.n_convs], 16, 16), dtype=torch.float32, device=y.device)
    _lib.check(lib.odehip_convstack_forward(ctypes.byref(desc), _ptr(y), _ptr(out), b, int(bool(negate)), _ptr(ws),
                                            ws.numel(), _stream()))
    return out


def odeint_fixed(stack, method, z0, t, save=False, negate=False):
    """Whole fixed-grid trajectory in one C-ABI call.  Returns (T,B,C,16,16); with save=True also the private
    workspace holding every saved activation (input of odeint_fixed_backward)."""
    require_device_tensor(z0, "y0")
    desc = stack.refresh()
    z0 = z0.contiguous()
    b, c = z0.shape[0], z0.shape[1]
    if z0.dim() != 4 or tuple(z0.shape[2:]) != (16, 16) or c != desc.channels[0]:
        raise ValueError(f"y0 must be (B,{desc.channels[0]},16,16) (got {tuple(z0.shape)})")
    t64 = [float(v) for v in t.detach().to("cpu", torch.float64).tolist()]
    n = len(t64)
    lib = _lib.load()
    m = _lib.METHODS[method]
    nbytes = lib.odehip_odeint_workspace_bytes(ctypes.byref(desc), b, n, m, int(save))
    if save:  # private: it must survive untouched until backward
        ws = alloc_workspace(nbytes, z0.device)
    else:
        ws = workspace(("odeint", b, n, m, tuple(desc.channels)), nbytes, z0.device)
    out = torch.empty((n, b, c, 16, 16), dtype=torch.float32, device=z0.device)
    tarr = (ctypes.c_double * n)(*t64)
    fmt = ctypes.c_int(0)
    _lib.check(lib.odehip_odeint_fixed(ctypes.byref(desc), m, _ptr(z0), tarr, n, b, _ptr(out), int(save), int(bool(negate)), _ptr(ws),
                                       ws.numel(), ctypes.byref(fmt), _stream()))
    if save:
        ws._odehip_saved_format = int(fmt.value)   # how the workspace holds the saved tensors (the backward call must be told)
    return (out, ws) if save else out


def odeint_fixed_backward(stack, method, t, batch, grad_out, ws):
    """Gradients of the discrete fixed-grid solver: (grad_z0, [grad_w...], [grad_b...])."""
    require_device_tensor(grad_out, "grad_out")
    desc = stack.refresh()
    dg = stack.dgrad_desc()
    grad_out = grad_out.contiguous()
    t64 = [float(v) for v in t.detach().to("cpu", torch.float64).tolist()]
    n = len(t64)
    c = desc.channels[0]
    gz0 = torch.empty((batch, c, 16, 16), dtype=torch.float32, device=grad_out.device)
    gws = [torch.empty_like(cv.weight) for cv in stack.convs]
    gbs = [torch.empty_like(cv.bias) for cv in stack.convs]
    nl = len(gws)
    gw_arr = (ctypes.c_void_p * nl)(*[g.data_ptr() for g in gws])
    gb_arr = (ctypes.c_void_p * nl)(*[g.data_ptr() for g in gbs])
    tarr = (ctypes.c_double * n)(*t64)
    _lib.check(_lib.load().odehip_odeint_fixed_backward(ctypes.byref(desc), ctypes.byref(dg), _lib.METHODS[method], tarr, n,
                                                        batch, _ptr(grad_out), _ptr(gz0), gw_arr, gb_arr,
                                                        int(getattr(ws, "_odehip_saved_format", 0)), _ptr(ws), ws.numel(), _stream()))
    return gz0, gws, gbs


LOG_CAP = 2048   # accepted steps the forward reports back (the backward pass re-integrates them)


def odeint_dopri5(stack, z0, t, rtol, atol, first_step=0.0, max_steps=0, negate=False):
    """Adaptive dopri5 trajectory; returns ((T,B,C,16,16), stats dict).  stats["accepted"] = [(t0, dt), ...]."""
    collect_pending_solves()   # an asynchronous solve still in flight owns the shared ("dopri5", ...) workspace
    require_device_tensor(z0, "y0")
    desc = stack.refresh()
    z0 = z0.contiguous()
    b, c = z0.shape[0], z0.shape[1]
    if z0.dim() != 4 or tuple(z0.shape[2:]) != (16, 16) or c != desc.channels[0]:
        raise ValueError(f"y0 must be (B,{desc.channels[0]},16,16) (got {tuple(z0.shape)})")
    t64 = [float(v) for v in t.detach().to("cpu", torch.float64).tolist()]
    n = len(t64)
    lib = _lib.load()
    nbytes = lib.odehip_dopri5_workspace_bytes(ctypes.byref(desc), b, n)
    ws = workspace(("dopri5", b, n, tuple(desc.channels)), nbytes, z0.device)
    out = torch.empty((n, b, c, 16, 16), dtype=torch.float32, device=z0.device)
    tarr = (ctypes.c_double * n)(*t64)
    stats = (ctypes.c_int * 4)()
    log = (ctypes.c_double * (2 * LOG_CAP))()
    _lib.check(lib.odehip_odeint_dopri5(ctypes.byref(desc), _ptr(z0), tarr, n, b, float(rtol), float(atol), float(first_step or 0.0), int(max_steps),
                                        int(bool(negate)), _ptr(out), stats, log, LOG_CAP, _ptr(ws), ws.numel(), _stream()))
    k = min(int(stats[1]), LOG_CAP)
    return out, {"nfe": stats[0], "n_accept": stats[1], "n_reject": stats[2], "attempts_enqueued": stats[3],
                 "accepted": [(log[2 * i], log[2 * i + 1]) for i in range(k)]}


_dopri5_save_slots = 8   # slots of a saving forward's workspace; doubled (up to 64) after a forward that accepted more steps

# ---- asynchronous dopri5 (include/odecgru_hip.h: odehip_odeint_dopri5_start / _collect) ------------------------------------------
# The synchronous entry points return when the device-side controller has reported completion, so the host cannot enqueue what comes
# BEHIND the solver meanwhile: in a whole training step (encoder -> solver -> decoder -> loss -> backward, ~600 launches) the device
# then idles while the backward pass is being enqueued (ODEConvGRU, B=64, dopri5: 20.9 ms per step against 12.1 ms of device work).
# With set_async_dopri5(True) a solve only ENQUEUES a few attempted steps and returns; its outcome (step counts, accepted-step log,
# whether activations were kept -- and any error: dt underflow, max_num_steps, non-finite state) is read when it is needed: at the
# backward pass, at the first look into ode_rl_amd.last_stats, or at the next dopri5 call.  Off by default: torchdiffeq raises such
# errors from odeint() itself, and so does the synchronous path.
_async_dopri5 = os.environ.get("ODEHIP_DOPRI5_ASYNC") == "1"
_async_attempts = 8      # attempted steps enqueued up front; follows what the previous solve needed (+ margin)
ASYNC_ATTEMPTS_MAX = 256
_pending_solves = []     # PendingDopri5 objects not collected yet


def set_async_dopri5(on=True):
    """Switch the asynchronous dopri5 forward on / off (see above); returns the previous setting."""
    global _async_dopri5
    was = _async_dopri5
    _async_dopri5 = bool(on)
    if not on:
        collect_pending_solves()   # may raise a pending solve's error; the switch is already off then
    return was


def collect_pending_solves():
    """Wait for every dopri5 solve started asynchronously and surface its error, if any."""
    while _pending_solves:
        _pending_solves[0].collect()


class PendingDopri5:
    """A dopri5 solve that has been enqueued but whose outcome has not been read yet."""

    def __init__(self, token, keep, slots, ws):
        self.token, self._keep, self.slots, self.ws = token, keep, slots, ws
        self._result = None
        _pending_solves.append(self)

    def collect(self):
        """(stats dict, saved) -- saved = (workspace, slots) if the activations of the accepted steps were kept, else None."""
        global _dopri5_save_slots, _async_attempts
        if isinstance(self._result, BaseException):   # the solve failed: every later look at it raises the same error
            raise self._result
        if self._result is None:
            if self in _pending_solves:
                _pending_solves.remove(self)
            stats = (ctypes.c_int * 4)()
            log = (ctypes.c_double * (2 * LOG_CAP))()
            saved = ctypes.c_int(0)
            lib = _lib.load()
            try:
                _lib.check(lib.odehip_odeint_dopri5_collect(int(self.token), stats, log, LOG_CAP, ctypes.byref(saved)))
            except _lib.AsyncSolveTruncated as e:
                # the attempts enqueued up front did not finish the solve and its consumers have read NaN frames: the NEXT solve gets
                # twice as many (the caller repeats its step; train_batch does)
                _async_attempts = min(ASYNC_ATTEMPTS_MAX, max(2 * int(stats[3]), _async_attempts))
                self._result = e
                raise
            except BaseException as e:
                self._result = e
                raise
            finally:
                self._keep = None
            k = min(int(stats[1]), LOG_CAP)
            if self.slots and int(stats[1]) > self.slots:
                _dopri5_save_slots = min(64, max(2 * self.slots, int(stats[1]) + 2))
            # what this solve needed plus a margin of a quarter (at least 2): an attempt queued behind `done` costs three empty
            # launches, an attempt too few costs the step (AsyncSolveTruncated).  Never below what the last solve was given unless
            # it used less than half of it -- the count drifts slowly as the dynamics train
            used = int(stats[1]) + int(stats[2])
            want = used + max(2, used // 4)
            _async_attempts = max(4, min(ASYNC_ATTEMPTS_MAX, want if want > _async_attempts or 2 * want < _async_attempts else _async_attempts))
            st = {"nfe": stats[0], "n_accept": stats[1], "n_reject": stats[2], "attempts_enqueued": stats[3],
                  "accepted": [(log[2 * i], log[2 * i + 1]) for i in range(k)], "saved": bool(saved.value)}
            self._result = (st, (self.ws, self.slots) if saved.value else None)
            self.ws = None
        return self._result


class LazyStats(dict):
    """`ode_rl_amd.last_stats`: a dict that, after an asynchronous solve, fills itself on first access."""
    _pending = None

    def _bind(self, pending):
        dict.clear(self)
        self._pending = pending

    def _resolve(self):
        p, self._pending = self._pending, None
        if p is not None:
            dict.update(self, p.collect()[0])

    def clear(self):
        self._pending = None
        dict.clear(self)

    def __getitem__(self, k):
        self._resolve()
        return dict.__getitem__(self, k)

    def get(self, k, d=None):
        self._resolve()
        return dict.get(self, k, d)

    def __iter__(self):
        self._resolve()
        return dict.__iter__(self)

    def __len__(self):
        self._resolve()
        return dict.__len__(self)

    def __contains__(self, k):
        self._resolve()
        return dict.__contains__(self, k)

    def keys(self):
        self._resolve()
        return dict.keys(self)

    def items(self):
        self._resolve()
        return dict.items(self)

    def values(self):
        self._resolve()
        return dict.values(self)

    def copy(self):
        self._resolve()
        return dict(dict.items(self))

    def __eq__(self, other):
        self._resolve()
        return dict.__eq__(self, other)

    def __repr__(self):
        self._resolve()
        return dict.__repr__(self)


def odeint_dopri5_start(stack, z0, t, rtol, atol, first_step=0.0, max_steps=0, save=False):
    """Asynchronous dopri5 forward: returns (out, PendingDopri5) at once.  save=True keeps the activations of the accepted steps
    (as odeint_dopri5_saving does) in a private workspace held by the pending object."""
    collect_pending_solves()      # one solve in flight per process: the shared workspaces and the library's slots are free again
    require_device_tensor(z0, "y0")
    desc = stack.refresh()
    z0 = z0.contiguous()
    b, c = z0.shape[0], z0.shape[1]
    if z0.dim() != 4 or tuple(z0.shape[2:]) != (16, 16) or c != desc.channels[0]:
        raise ValueError(f"y0 must be (B,{desc.channels[0]},16,16) (got {tuple(z0.shape)})")
    t64 = [float(v) for v in t.detach().to("cpu", torch.float64).tolist()]
    n = len(t64)
    lib = _lib.load()
    slots = _dopri5_save_slots if save and os.environ.get("ODEHIP_DOPRI5_SAVE") != "0" else 0
    if slots:
        ws = alloc_workspace(lib.odehip_dopri5_saving_workspace_bytes(ctypes.byref(desc), b, n, slots), z0.device)
    else:
        ws = workspace(("dopri5", b, n, tuple(desc.channels)), lib.odehip_dopri5_workspace_bytes(ctypes.byref(desc), b, n), z0.device)
    out = torch.empty((n, b, c, 16, 16), dtype=torch.float32, device=z0.device)
    tarr = (ctypes.c_double * n)(*t64)
    token = ctypes.c_int(-1)
    _lib.check(lib.odehip_odeint_dopri5_start(ctypes.byref(desc), _ptr(z0), tarr, n, b, float(rtol), float(atol), float(first_step or 0.0),
                                              int(max_steps), _ptr(out), int(slots), int(_async_attempts), ctypes.byref(token), _ptr(ws),
                                              ws.numel(), _stream()))
    return out, PendingDopri5(token.value, (z0, out, ws), slots, ws if slots else None)


def odeint_dopri5_saving(stack, z0, t, rtol, atol, first_step=0.0, max_steps=0):
    """The forward of a dopri5 TRAINING step: odeint_dopri5 that also keeps the stage inputs and hidden activations of every accepted
    step in a private workspace, so that the backward pass needs no re-integration.  Returns (out, stats, saved) with saved =
    (workspace, max_accept), or None when nothing was kept (other stacks than 64-channel fp32, persistent walk off, more accepted
    steps than slots): the caller then takes odeint_dopri5_backward."""
    global _dopri5_save_slots
    if _async_dopri5:   # enqueue only: stats and `saved` are read later (LazyStats / PendingDopri5.collect())
        out, pending = odeint_dopri5_start(stack, z0, t, rtol, atol, first_step=first_step, max_steps=max_steps, save=True)
        return out, pending, pending
    if os.environ.get("ODEHIP_DOPRI5_SAVE") == "0":   # A/B switch: always re-integrate in the backward pass
        out, st = odeint_dopri5(stack, z0, t, rtol, atol, first_step=first_step, max_steps=max_steps)
        st["saved"] = False
        return out, st, None
    collect_pending_solves()
    require_device_tensor(z0, "y0")
    desc = stack.refresh()
    z0 = z0.contiguous()
    b, c = z0.shape[0], z0.shape[1]
    if z0.dim() != 4 or tuple(z0.shape[2:]) != (16, 16) or c != desc.channels[0]:
        raise ValueError(f"y0 must be (B,{desc.channels[0]},16,16) (got {tuple(z0.shape)})")
    t64 = [float(v) for v in t.detach().to("cpu", torch.float64).tolist()]
    n = len(t64)
    lib = _lib.load()
    slots = _dopri5_save_slots
    nbytes = lib.odehip_dopri5_saving_workspace_bytes(ctypes.byref(desc), b, n, slots)
    ws = alloc_workspace(nbytes, z0.device)   # private: it must survive untouched until backward
    out = torch.empty((n, b, c, 16, 16), dtype=torch.float32, device=z0.device)
    tarr = (ctypes.c_double * n)(*t64)
    stats = (ctypes.c_int * 4)()
    log = (ctypes.c_double * (2 * LOG_CAP))()
    saved = ctypes.c_int(0)
    _lib.check(lib.odehip_odeint_dopri5_saving(ctypes.byref(desc), _ptr(z0), tarr, n, b, float(rtol), float(atol), float(first_step or 0.0),
                                               int(max_steps), _ptr(out), stats, log, LOG_CAP, slots, ctypes.byref(saved), _ptr(ws), ws.numel(),
                                               _stream()))
    k = min(int(stats[1]), LOG_CAP)
    if int(stats[1]) > slots:
        _dopri5_save_slots = min(64, max(2 * slots, int(stats[1]) + 2))
    st = {"nfe": stats[0], "n_accept": stats[1], "n_reject": stats[2], "attempts_enqueued": stats[3],
          "accepted": [(log[2 * i], log[2 * i + 1]) for i in range(k)], "saved": bool(saved.value)}
    return out, st, ((ws, slots) if saved.value else None)


def odeint_dopri5_backward_saved(stack, t, accepted, grad_out, saved):
    """Backward of odeint_dopri5_saving: the reverse sweep over the kept activations (no re-integration)."""
    require_device_tensor(grad_out, "grad_out")
    ws, slots = saved
    desc = stack.refresh()
    dg = stack.dgrad_desc()
    grad_out = grad_out.contiguous()
    t64 = [float(v) for v in t.detach().to("cpu", torch.float64).tolist()]
    n, b, c = len(t64), grad_out.shape[1], desc.channels[0]
    lib = _lib.load()
    gz0 = torch.empty((b, c, 16, 16), dtype=torch.float32, device=grad_out.device)
    gws = [torch.empty_like(cv.weight) for cv in stack.convs]
    gbs = [torch.empty_like(cv.bias) for cv in stack.convs]
    nl = len(gws)
    gw_arr = (ctypes.c_void_p * nl)(*[g.data_ptr() for g in gws])
    gb_arr = (ctypes.c_void_p * nl)(*[g.data_ptr() for g in gbs])
    tarr = (ctypes.c_double * n)(*t64)
    flat = [v for pair in accepted for v in pair]
    larr = (ctypes.c_double * max(len(flat), 1))(*flat)
    _lib.check(lib.odehip_odeint_dopri5_backward_saved(ctypes.byref(desc), ctypes.byref(dg), tarr, n, b, larr, len(accepted), _ptr(grad_out),
                                                       _ptr(gz0), gw_arr, gb_arr, int(slots), _ptr(ws), ws.numel(), _stream()))
    return gz0, gws, gbs


def odeint_dopri5_backward(stack, t, accepted, z0, grad_out):
    """Gradient of the accepted dopri5 steps (what autograd through torchdiffeq computes): (grad_z0, [grad_w], [grad_b])."""
    require_device_tensor(grad_out, "grad_out")
    require_device_tensor(z0, "y0")
    desc = stack.refresh()
    dg = stack.dgrad_desc()
    grad_out, z0 = grad_out.contiguous(), z0.contiguous()
    t64 = [float(v) for v in t.detach().to("cpu", torch.float64).tolist()]
    n, b, c = len(t64), z0.shape[0], desc.channels[0]
    ns = len(accepted)
    lib = _lib.load()
    nbytes = lib.odehip_dopri5_backward_workspace_bytes(ctypes.byref(desc), b, n, ns)
    # keyed WITHOUT the accepted-step count: that count drifts as the weights train, and a key per count would leave one
    # multi-GB buffer behind for every count ever seen; workspace() replaces a buffer that has become too small
    ws = workspace(("dopri5_bwd", b, n, tuple(desc.channels)), nbytes, grad_out.device)
    gz0 = torch.empty((b, c, 16, 16), dtype=torch.float32, device=grad_out.device)
    gws = [torch.empty_like(cv.weight) for cv in stack.convs]
    gbs = [torch.empty_like(cv.bias) for cv in stack.convs]
    nl = len(gws)
    gw_arr = (ctypes.c_void_p * nl)(*[g.data_ptr() for g in gws])
    gb_arr = (ctypes.c_void_p * nl)(*[g.data_ptr() for g in gbs])
    tarr = (ctypes.c_double * n)(*t64)
    flat = [v for pair in accepted for v in pair]
    larr = (ctypes.c_double * max(len(flat), 1))(*flat)
    _lib.check(lib.odehip_odeint_dopri5_backward(ctypes.byref(desc), ctypes.byref(dg), tarr, n, b, larr, ns, _ptr(z0),
                                                 _ptr(grad_out), _ptr(gz0), gw_arr, gb_arr, _ptr(ws), ws.numel(), _stream()))
    return gz0, gws, gbs


def odeint_adjoint_backward(stack, method, t, y_traj, grad_out):
    """torchdiffeq odeint_adjoint backward for the fixed-grid methods: (grad_z0, [grad_w...], [grad_b...])."""
    require_device_tensor(grad_out, "grad_out")
    require_device_tensor(y_traj, "y_traj")
    desc = stack.refresh()
    dg = stack.dgrad_desc()
    grad_out, y_traj = grad_out.contiguous(), y_traj.contiguous()
    t64 = [float(v) for v in t.detach().to("cpu", torch.float64).tolist()]
    n, b, c = len(t64), y_traj.shape[1], desc.channels[0]
    lib = _lib.load()
    m = _lib.METHODS[method]
    nbytes = lib.odehip_odeint_workspace_bytes(ctypes.byref(desc), b, n, m, 1)
    ws = workspace(("adjoint", b, n, m, tuple(desc.channels)), nbytes, grad_out.device)
    gz0 = torch.empty((b, c, 16, 16), dtype=torch.float32, device=grad_out.device)
    gws = [torch.empty_like(cv.weight) for cv in stack.convs]
    gbs = [torch.empty_like(cv.bias) for cv in stack.convs]
    nl = len(gws)
    gw_arr = (ctypes.c_void_p * nl)(*[g.data_ptr() for g in gws])
    gb_arr = (ctypes.c_void_p * nl)(*[g.data_ptr() for g in gbs])
    tarr = (ctypes.c_double * n)(*t64)
    _lib.check(lib.odehip_odeint_adjoint_backward(ctypes.byref(desc), ctypes.byref(dg), m, tarr, n, b, _ptr(y_traj),
                                                  _ptr(grad_out), _ptr(gz0), gw_arr, gb_arr, _ptr(ws), ws.numel(), _stream()))
    return gz0, gws, gbs


def odeint_adjoint_dopri5_backward(stack, t, y_traj, grad_out, rtol, atol, max_accept=None, stats=None, mixed_norm=False):
    """torchdiffeq odeint_adjoint backward with method="dopri5" (seminorm, or the default mixed norm): (grad_z0, [grad_w...], [grad_b...]).
    `stats`, if a dict, receives nfe / n_accept / n_reject of the backward solve."""
    collect_pending_solves()   # in particular the forward solve whose trajectory this integrates from (its error surfaces here)
    require_device_tensor(grad_out, "grad_out")
    require_device_tensor(y_traj, "y_traj")
    desc = stack.refresh()
    dg = stack.dgrad_desc()
    grad_out, y_traj = grad_out.contiguous(), y_traj.contiguous()
    t64 = [float(v) for v in t.detach().to("cpu", torch.float64).tolist()]
    n, b, c = len(t64), y_traj.shape[1], desc.channels[0]
    lib = _lib.load()
    if max_accept is None:  # every accepted step keeps its activations: default to what fits in 32 GiB, at most 256 steps
        per_step = (lib.odehip_adjoint_dopri5_workspace_bytes(ctypes.byref(desc), b, n, 2)
                    - lib.odehip_adjoint_dopri5_workspace_bytes(ctypes.byref(desc), b, n, 1))
        max_accept = int(max(4 * n, min(256, (32 << 30) // max(per_step, 1))))
    nbytes = lib.odehip_adjoint_dopri5_workspace_bytes(ctypes.byref(desc), b, n, int(max_accept))
    ws = workspace(("adjoint_dopri5", b, n, int(max_accept), tuple(desc.channels)), nbytes, grad_out.device)
    gz0 = torch.empty((b, c, 16, 16), dtype=torch.float32, device=grad_out.device)
    gws = [torch.empty_like(cv.weight) for cv in stack.convs]
    gbs = [torch.empty_like(cv.bias) for cv in stack.convs]
    nl = len(gws)
    gw_arr = (ctypes.c_void_p * nl)(*[g.data_ptr() for g in gws])
    gb_arr = (ctypes.c_void_p * nl)(*[g.data_ptr() for g in gbs])
    tarr = (ctypes.c_double * n)(*t64)
    st = (ctypes.c_int * 3)()
    _lib.check(lib.odehip_odeint_adjoint_dopri5_backward(ctypes.byref(desc), ctypes.byref(dg), tarr, n, b, float(rtol),
                                                         float(atol), _ptr(y_traj), _ptr(grad_out), _ptr(gz0), gw_arr, gb_arr,
                                                         int(max_accept), int(bool(mixed_norm)), st, _ptr(ws), ws.numel(), _stream()))
    if stats is not None:
        stats.update(nfe=int(st[0]), n_accept=int(st[1]), n_reject=int(st[2]))
    return gz0, gws, gbs


class PackedCell:
    """Packed parameters of a ConvGRUCell (conv_gates / conv_can Sequentials), refreshed on parameter change; one cache entry
    per compute dtype (the bf16 entry adds the bf16 weight images of the two 5x5 convs)."""

    def __init__(self, cell):
        self.cell = cell
        self._cache = {}
        self._stamp = None     # identity of the entry returned last (keys the derived caches)
        self.desc = None

    def _params(self):
        c = self.cell
        return [c.conv_gates[0].weight, c.conv_gates[0].bias, c.conv_gates[1].weight, c.conv_gates[1].bias,
                c.conv_can[0].weight, c.conv_can[0].bias, c.conv_can[1].weight, c.conv_can[1].bias]

    def refresh(self, mode=None):
        mode = mode or current_compute_dtype()
        ps = self._params()
        stamp = (mode, winograd5_enabled()) + tuple((p.data_ptr(), p._version) for p in ps)
        ent = self._cache.get(mode)
        if ent is not None and ent[0] == stamp:
            self._stamp, self.desc = ent[0], ent[1]
            return ent[1]
        for p in ps:
            require_device_tensor(p, "ConvGRUCell parameter")
        c = self.cell
        ks = c.conv_gates[0].kernel_size[0]
        if c.conv_gates[0].padding != (ks // 2, ks // 2):
            raise ValueError("the HIP ConvGRU supports 'same' padding only")
        if c.conv_gates[1].num_groups * 32 != 2 * c.hidden_dim or c.conv_can[1].num_groups * 32 != c.hidden_dim:
            raise ValueError("the HIP ConvGRU needs GroupNorm groups of 32 channels (hidden_dim multiple of 32)")
        want_wino = mode == "f32" and ks == 5 and winograd5_enabled() and c.input_channels % 8 == 0 and c.hidden_dim % 32 == 0
        packs = pack_conv_weights_many([(ps[0], 0, False), (ps[4], 0, False)] + ([(ps[0], 2, False), (ps[4], 2, False)] if want_wino else []))
        keep = [packs[0], ps[1].detach().contiguous(), ps[2].detach().contiguous(), ps[3].detach().contiguous(),
                packs[1], ps[5].detach().contiguous(), ps[6].detach().contiguous(), ps[7].detach().contiguous()]
        bf = [None, None]
        if mode == "bf16" and _bf16_cell_ok(c.input_channels, c.hidden_dim, ks):
            bf = [pack_conv_weight_bf16_ks(ps[0]), pack_conv_weight_bf16_ks(ps[4])]
        wino = packs[2:4] if want_wino else [None, None]
        keep = keep + wino
        d = _lib.ConvGRUCellDesc(input=c.input_channels, hidden=c.hidden_dim, ks=ks,
                                 w_gates_wino=wino[0].data_ptr() if wino[0] is not None else None,
                                 w_can_wino=wino[1].data_ptr() if wino[1] is not None else None,
                                 w_gates=keep[0].data_ptr(), b_gates=keep[1].data_ptr(), gn_gates_w=keep[2].data_ptr(),
                                 gn_gates_b=keep[3].data_ptr(), w_can=keep[4].data_ptr(), b_can=keep[5].data_ptr(),
                                 gn_can_w=keep[6].data_ptr(), gn_can_b=keep[7].data_ptr(),
                                 w_gates_bf16=bf[0].data_ptr() if bf[0] is not None else None,
                                 w_can_bf16=bf[1].data_ptr() if bf[1] is not None else None)
        self._cache[mode] = (stamp, d, keep, bf)
        self._stamp, self.desc = stamp, d
        return d


def _cell_bwd_packs(cell, d, mode, extra_jobs=()):
    """Transposed + flipped slices of the two 5x5 weights (frame half, state half), fp32 images and, in bf16 mode, bf16 ones; extra_jobs
    (pack_conv_weights_many tuples) ride in the same launch and their results follow the four slices in the first list."""
    i = d.input
    wg, wc = cell.conv_gates[0].weight.detach(), cell.conv_can[0].weight.detach()
    slices = [wg[:, :i], wg[:, i:], wc[:, :i], wc[:, i:]]
    slices = [w.contiguous() for w in slices]
    want_wino = mode == "f32" and d.ks == 5 and winograd5_enabled() and all(w.shape[0] % 8 == 0 and w.shape[1] % 32 == 0 for w in slices)
    packs = pack_conv_weights_many([(w, 0, True) for w in slices] + list(extra_jobs) + ([(w, 2, True) for w in slices] if want_wino else []))
    keep, extra = packs[:4], packs[4:4 + len(extra_jobs)]
    bf = [None] * 4
    if mode == "bf16" and d.ks == 5 and all(w.shape[0] <= 128 and w.shape[0] % 16 == 0 and w.shape[1] % 32 == 0 for w in slices):
        bf = [pack_conv_weight_bf16_ks(w, True) for w in slices]
    wino = packs[4 + len(extra_jobs):] if want_wino else [None] * 4
    return keep + extra, bf, wino


def convgru_cell_forward(packed_cell, x, h):
    require_device_tensor(x, "input_tensor")
    require_device_tensor(h, "h_cur")
    d = packed_cell.refresh()
    x, h = x.contiguous(), h.contiguous()
    b = x.shape[0]
    if tuple(x.shape) != (b, d.input, 16, 16) or tuple(h.shape) != (b, d.hidden, 16, 16):
        raise ValueError(f"ConvGRU step needs x (B,{d.input},16,16) and h (B,{d.hidden},16,16); got {tuple(x.shape)}, {tuple(h.shape)}")
    lib = _lib.load()
    nbytes = lib.odehip_convgru_cell_workspace_bytes(ctypes.byref(d), b)
    ws = workspace(("cgru", b, d.input, d.hidden), nbytes, x.device)
    out = torch.empty_like(h)
    _lib.check(lib.odehip_convgru_cell_forward(ctypes.byref(d), _ptr(x), _ptr(h), _ptr(out), b, _ptr(ws), ws.numel(), _stream()))
    return out


def convgru_cell_backward(packed_cell, x, h, grad_h_next):
    """Backward of one ConvGRU step: (grad_x, grad_h, [gradients of packed_cell._params()])."""
    for t_, n_ in ((x, "input_tensor"), (h, "h_cur"), (grad_h_next, "grad_h_next")):
        require_device_tensor(t_, n_)
    d = packed_cell.refresh()
    cached = getattr(packed_cell, "_bwd", None)
    if cached is None or cached[0] != packed_cell._stamp:
        keep, bf, wino = _cell_bwd_packs(packed_cell.cell, d, current_compute_dtype())
        bw = _lib.ConvGRUCellBwd(*[k.data_ptr() for k in keep])
        for j in range(4):
            bw.bf16[j] = bf[j].data_ptr() if bf[j] is not None else None
            bw.wino[j] = wino[j].data_ptr() if wino[j] is not None else None
        cached = packed_cell._bwd = (packed_cell._stamp, bw, keep, bf, wino)
    bw = cached[1]
    x, h, grad_h_next = x.contiguous(), h.contiguous(), grad_h_next.contiguous()
    b = x.shape[0]
    params = packed_cell._params()
    grads = [torch.empty_like(p) for p in params]
    g = _lib.ConvGRUCellGrads(*[t_.data_ptr() for t_ in grads])
    gx, gh = torch.empty_like(x), torch.empty_like(h)
    lib = _lib.load()
    nbytes = lib.odehip_convgru_cell_backward_workspace_bytes(ctypes.byref(d), b)
    ws = workspace(("cgru_bwd", b, d.input, d.hidden), nbytes, x.device)
    _lib.check(lib.odehip_convgru_cell_backward(ctypes.byref(d), ctypes.byref(bw), _ptr(x), _ptr(h), _ptr(grad_h_next), _ptr(gx),
                                                _ptr(gh), ctypes.byref(g), b, _ptr(ws), ws.numel(), _stream()))
    return gx, gh, grads


class PackedEncoder:
    """Everything `ODEConvGRUCell.forward` needs on the device: encoder dynamics, cell, 1x1 head."""

    def __init__(self, f_stack, packed_cell, head):
        self.f_stack, self.packed_cell, self.head = f_stack, packed_cell, head
        self._stamp = None
        self._keep = None
        self.desc = None

    def refresh(self):
        fd = self.f_stack.refresh()
        cd = self.packed_cell.refresh()
        h0, h1 = self.head[0], self.head[2]
        ps = [h0.weight, h0.bias, h1.weight, h1.bias]
        stamp = (id(fd), id(cd)) + tuple((p.data_ptr(), p._version) for p in ps)
        if stamp == self._stamp:
            return self.desc
        hw = pack_conv_weights_many([(h0.weight, 0, False), (h1.weight, 0, False)])
        keep = [hw[0], h0.bias.detach().contiguous(), hw[1], h1.bias.detach().contiguous()]
        d = _lib.EncoderDesc()
        d.f_enc = fd
        d.cell = cd
        d.head_hidden = h0.out_channels
        d.out_ch = h1.out_channels // 2
        d.w_head0, d.b_head0, d.w_head1, d.b_head1 = (k.data_ptr() for k in keep)
        self._keep, self.desc, self._stamp = keep, d, stamp
        return d


def encoder_params(enc):
    """Parameters of the encoder in the order the training path returns their gradients."""
    cell = enc.packed_cell.cell
    ps = []
    for c in enc.f_stack.convs:
        ps += [c.weight, c.bias]
    ps += [cell.conv_gates[0].weight, cell.conv_gates[0].bias, cell.conv_gates[1].weight, cell.conv_gates[1].bias,
           cell.conv_can[0].weight, cell.conv_can[0].bias, cell.conv_can[1].weight, cell.conv_can[1].bias,
           enc.head[0].weight, enc.head[0].bias, enc.head[2].weight, enc.head[2].bias]
    return ps


def _encoder_bwd_desc(enc):
    """Transposed + flipped weight images of the input-gradient convs (rebuilt when a parameter changes)."""
    d = enc.refresh()
    stamp = enc._stamp
    cached = getattr(enc, "_bwd", None)
    if cached is not None and cached[0] is stamp:
        return cached[1]
    keep, bf, wino = _cell_bwd_packs(enc.packed_cell.cell, d.cell, current_compute_dtype(),
                                     extra_jobs=[(enc.head[0].weight, 0, True), (enc.head[2].weight, 0, True)])
    b = _lib.EncoderBwd()
    b.f_dgrad = enc.f_stack.dgrad_desc()
    b.w_gates_dx, b.w_gates_dh, b.w_can_dx, b.w_can_dh, b.w_head0_t, b.w_head1_t = (k.data_ptr() for k in keep)
    for j in range(4):
        b.bf16[j] = bf[j].data_ptr() if bf[j] is not None else None
        b.wino[j] = wino[j].data_ptr() if wino[j] is not None else None
    keep = keep + bf + wino
    enc._bwd = (stamp, b, keep)
    return b


def odeconvgru_encode_train(enc, inputs, timesteps, want_latent=False, run_backwards=True):
    """Forward of the training path: returns (mean, std, latent_ys or None, saved) -- `saved` holds the workspace the backward
    call needs."""
    require_device_tensor(inputs, "inputs")
    d = enc.refresh()
    inputs = inputs.contiguous()
    t, b, c = inputs.shape[0], inputs.shape[1], inputs.shape[2]
    if inputs.dim() != 5 or tuple(inputs.shape[3:]) != (16, 16) or c != d.cell.hidden:
        raise ValueError(f"inputs must be (T,B,{d.cell.hidden},16,16) time-first (got {tuple(inputs.shape)})")
    t64 = [float(v) for v in host_times(timesteps).tolist()]
    assert t == len(t64), "Sequence length should be same as time_steps"
    lib = _lib.load()
    nbytes = lib.odehip_encoder_train_workspace_bytes(ctypes.byref(d), t, b)
    ws = alloc_workspace(nbytes, inputs.device)   # owned by this call's graph node
    mean = torch.empty((b, d.out_ch, 16, 16), dtype=torch.float32, device=inputs.device)
    std = torch.empty_like(mean)
    latent = torch.empty((b, t, c, 16, 16), dtype=torch.float32, device=inputs.device) if want_latent else None
    tarr = (ctypes.c_double * t)(*t64)
    _lib.check(lib.odehip_odeconvgru_encode_train(ctypes.byref(d), _ptr(inputs), tarr, t, b, int(bool(run_backwards)), _ptr(mean),
                                                  _ptr(std), _ptr(latent), _ptr(ws), ws.numel(), _stream()))
    return mean, std, latent, (ws, t64, t, b, c, int(bool(run_backwards)))


def odeconvgru_encode_backward(enc, saved, grad_mean, grad_std, grad_latent=None):
    """(grad_inputs (T,B,C,16,16), [gradient of every tensor of encoder_params(enc)]); grad_latent: what arrives through latent_ys."""
    ws, t64, t, b, c, run_backwards = saved
    d = enc.refresh()
    bw = _encoder_bwd_desc(enc)
    dev = ws.device
    grad_mean = (torch.zeros((b, d.out_ch, 16, 16), device=dev) if grad_mean is None else grad_mean).contiguous()
    grad_std = (torch.zeros((b, d.out_ch, 16, 16), device=dev) if grad_std is None else grad_std).contiguous()
    require_device_tensor(grad_mean, "grad_mean")
    require_device_tensor(grad_std, "grad_std")
    if grad_latent is not None:
        grad_latent = grad_latent.contiguous()
        require_device_tensor(grad_latent, "grad_latent")
        if tuple(grad_latent.shape) != (b, t, c, 16, 16):
            raise ValueError(f"grad_latent must be ({b},{t},{c},16,16), got {tuple(grad_latent.shape)}")
    params = encoder_params(enc)
    grads = [torch.empty_like(p) for p in params]
    g = _lib.EncoderGrads()
    nl = len(enc.f_stack.convs)
    for l in range(nl):
        g.f_w[l] = grads[2 * l].data_ptr()
        g.f_b[l] = grads[2 * l + 1].data_ptr()
    (g.w_gates, g.b_gates, g.gn_gates_w, g.gn_gates_b, g.w_can, g.b_can, g.gn_can_w, g.gn_can_b, g.w_head0, g.b_head0, g.w_head1,
     g.b_head1) = (x.data_ptr() for x in grads[2 * nl:])
    gin = torch.empty((t, b, c, 16, 16), dtype=torch.float32, device=dev)
    tarr = (ctypes.c_double * t)(*t64)
    _lib.check(_lib.load().odehip_odeconvgru_encode_backward(ctypes.byref(d), ctypes.byref(bw), tarr, t, b, run_backwards,
                                                             _ptr(grad_mean), _ptr(grad_std), _ptr(grad_latent), _ptr(gin),
                                                             ctypes.byref(g), _ptr(ws), ws.numel(), _stream()))
    return gin, grads


def odeconvgru_encode(enc, inputs, timesteps, want_latent=False, run_backwards=True):
    require_device_tensor(inputs, "inputs")
    d = enc.refresh()
    inputs = inputs.contiguous()
    t, b, c = inputs.shape[0], inputs.shape[1], inputs.shape[2]
    if inputs.dim() != 5 or tuple(inputs.shape[3:]) != (16, 16) or c != d.cell.hidden:
        raise ValueError(f"inputs must be (T,B,{d.cell.hidden},16,16) time-first (got {tuple(inputs.shape)})")
    t64 = [float(v) for v in host_times(timesteps).tolist()]
    assert t == len(t64), "Sequence length should be same as time_steps"
    lib = _lib.load()
    nbytes = lib.odehip_encoder_workspace_bytes(ctypes.byref(d), t, b)
    ws = workspace(("enc", t, b, c), nbytes, inputs.device)
    mean = torch.empty((b, d.out_ch, 16, 16), dtype=torch.float32, device=inputs.device)
    std = torch.empty_like(mean)
    latent = torch.empty((b, t, c, 16, 16), dtype=torch.float32, device=inputs.device) if want_latent else None
    tarr = (ctypes.c_double * t)(*t64)
    _lib.check(lib.odehip_odeconvgru_encode(ctypes.byref(d), _ptr(inputs), tarr, t, b, int(bool(run_backwards)), _ptr(mean), _ptr(std),
                                            _ptr(latent), _ptr(ws), ws.numel(), _stream()))
    return mean, std, latent


# ---- VidODE's warp chain + mask compositing (csrc/warp.hip) ---------------------------------------------------------------
def warp_composite(pred_outputs, start_image, grid_x, grid_y):
    """pred_outputs (B,T,c+3,H,W), start_image (B,c,H,W) -> pred_x, warped (B,T,c,H,W), masks (B,T,1,H,W); one launch."""
    for t_, n_ in ((pred_outputs, "pred_outputs"), (start_image, "start_image"), (grid_x, "grid_x"), (grid_y, "grid_y")):
        require_device_tensor(t_, n_)
    pred_outputs, start_image = pred_outputs.contiguous(), start_image.contiguous()
    b, t, cc, h, w = pred_outputs.shape
    c = cc - 3
    if tuple(start_image.shape) != (b, c, h, w) or grid_x.numel() != w or grid_y.numel() != h:
        raise ValueError(f"warp_composite: start_image must be (B,{c},{h},{w}) and the grids ({w},), ({h},); got {tuple(start_image.shape)}")
    pred_x = torch.empty((b, t, c, h, w), dtype=torch.float32, device=pred_outputs.device)
    warped = torch.empty_like(pred_x)
    masks = torch.empty((b, t, 1, h, w), dtype=torch.float32, device=pred_outputs.device)
    _lib.check(_lib.load().odehip_warp_composite(_ptr(pred_outputs), _ptr(start_image), _ptr(grid_x.contiguous()), _ptr(grid_y.contiguous()),
                                                 b, t, c, h, w, _ptr(pred_x), _ptr(warped), _ptr(masks), _stream()))
    return pred_x, warped, masks


def warp_composite_backward(pred_outputs, start_image, warped, grid_x, grid_y, g_pred_x, g_warped, g_masks, want_start_grad):
    b, t, cc, h, w = pred_outputs.shape
    c = cc - 3
    g_pred_x = (torch.zeros_like(warped) if g_pred_x is None else g_pred_x).contiguous()
    g_warped = g_warped.contiguous() if g_warped is not None else None
    g_masks = g_masks.contiguous() if g_masks is not None else None
    for t_ in (g_pred_x, g_warped, g_masks):
        if t_ is not None:
            require_device_tensor(t_, "gradient")
    g_po = torch.empty_like(pred_outputs)
    g_start = torch.empty_like(start_image) if want_start_grad else None
    _lib.check(_lib.load().odehip_warp_composite_backward(_ptr(pred_outputs), _ptr(start_image), _ptr(warped), _ptr(grid_x), _ptr(grid_y),
                                                          _ptr(g_pred_x), _ptr(g_warped), _ptr(g_masks), b, t, c, h, w, _ptr(g_po),
                                                          _ptr(g_start), _stream()))
    return g_po, g_start


def upsample2x(x):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False) of an (..., H, W) fp32 device tensor (csrc/upsample.hip)."""
    require_device_tensor(x, "input")
    if x.dim() < 2 or x.shape[-1] % 2:
        raise ValueError(f"upsample2x: needs (..., H, W) with an even W, got {tuple(x.shape)}")
    x = x.contiguous()
    h, w = x.shape[-2:]
    out = torch.empty(tuple(x.shape[:-2]) + (2 * h, 2 * w), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().odehip_upsample2x_bilinear(_ptr(x), _ptr(out), x.numel() // (h * w), h, w, _stream()))
    return out


def upsample2x_backward(g):
    require_device_tensor(g, "gradient")
    g = g.contiguous()
    if g.dim() < 2 or g.shape[-2] % 2 or g.shape[-1] % 2:
        raise ValueError(f"upsample2x backward: needs (..., 2H, 2W), got {tuple(g.shape)}")
    h2, w2 = g.shape[-2:]
    gin = torch.empty(tuple(g.shape[:-2]) + (h2 // 2, w2 // 2), dtype=torch.float32, device=g.device)
    _lib.check(_lib.load().odehip_upsample2x_bilinear_backward(_ptr(g), _ptr(gin), gin.numel() // ((h2 // 2) * (w2 // 2)), h2 // 2, w2 // 2, _stream()))
    return gin


def bn_relu_up_forward(x, bn, upsample, conv_bias=None):
    """relu(bn(x)) [upsampled x2] in one pass (csrc/bn_relu_up.hip); bn: an nn.BatchNorm2d (its running statistics are updated in
    train() mode exactly as the module would).  Returns (out, saved) with saved = (mean, invstd, scale, shift) for the backward.
    conv_bias: the bias of the convolution that produced x, NOT added to x (the caller ran the convolution without it).  A
    per-channel constant in front of BatchNorm cancels in train() mode -- only running_mean sees it -- and shifts the running mean in
    eval() mode: folding it here saves the bias-add pass over x and, in the backward, the reduction of a bias gradient that is exactly
    zero in train() mode (rocprofv3 of a VidODE step: 26 such convolutions, a 17-30 us add and a 40 us reduction each)."""
    require_device_tensor(x, "input")
    if x.dim() != 4 or x.shape[3] % 4:
        raise ValueError(f"bn_relu_up: needs (N, C, H, W) with W % 4 == 0, got {tuple(x.shape)}")
    x = x.contiguous()
    n, c, h, w = x.shape
    dev = x.device
    if bn.num_features != c or type(bn.momentum) is not float:   # (momentum=None -- a cumulative average -- is not built)
        raise ValueError(f"bn_relu_up: BatchNorm2d({bn.num_features}, momentum={bn.momentum}) on a {c}-channel input")
    for name in ("weight", "bias", "running_mean", "running_var"):
        p = getattr(bn, name)
        if p is not None:
            require_device_tensor(p, f"BatchNorm2d.{name}")
            if p.device != dev or p.numel() != c or not p.is_contiguous():
                raise ValueError(f"bn_relu_up: BatchNorm2d.{name} must be a contiguous ({c},) tensor on {dev}")
    if conv_bias is not None:
        require_device_tensor(conv_bias, "conv_bias")
        if conv_bias.device != dev or conv_bias.numel() != c or not conv_bias.is_contiguous():
            raise ValueError(f"bn_relu_up: conv_bias must be a contiguous ({c},) tensor on {dev}")
        conv_bias = conv_bias.detach()
    training = bn.training or bn.running_mean is None
    tracked = bn.track_running_stats and bn.running_mean is not None and bn.running_var is not None
    run_mean = bn.running_mean if tracked else None
    if conv_bias is not None and not training:
        run_mean = bn.running_mean - conv_bias      # (x + b - mean) = (x - (mean - b)); mean_out = this, so the backward's xhat agrees
    stats = torch.empty((4, c), dtype=torch.float32, device=dev)
    out = torch.empty((n, c, 2 * h, 2 * w) if upsample else (n, c, h, w), dtype=torch.float32, device=dev)
    lib = _lib.load()
    nws = lib.odehip_bn_workspace_bytes(c)
    ws = workspace(("bn", c), nws + 8 * c, dev)
    if training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    _lib.check(lib.odehip_bn_relu_up2x_forward(_ptr(x), n, c, h, w, _ptr(bn.weight), _ptr(bn.bias), _ptr(run_mean),
                                               _ptr(bn.running_var) if tracked else None,
                                               int(training), float(bn.momentum), float(bn.eps), int(bool(upsample)), _ptr(out),
                                               _ptr(stats[0]), _ptr(stats[1]), _ptr(stats[2]), _ptr(stats[3]), _ptr(ws), ws.numel(), _stream()))
    if conv_bias is not None and training and tracked:
        bn.running_mean.add_(conv_bias, alpha=bn.momentum)   # the mean of (x + b) is mean(x) + b
    return out, (stats, training)


def bn_relu_up_backward(grad_out, x, saved, upsample):
    stats, training = saved
    require_device_tensor(grad_out, "gradient")
    grad_out, x = grad_out.contiguous(), x.contiguous()
    n, c, h, w = x.shape
    if tuple(grad_out.shape) != ((n, c, 2 * h, 2 * w) if upsample else (n, c, h, w)):
        raise ValueError(f"bn_relu_up backward: gradient of shape {tuple(grad_out.shape)} for an input of shape {tuple(x.shape)} (upsample={bool(upsample)})")
    dev = x.device
    lib = _lib.load()
    gx = torch.empty_like(x)
    g_pre = torch.empty_like(x)
    gw = torch.empty(c, dtype=torch.float32, device=dev)
    gb = torch.empty(c, dtype=torch.float32, device=dev)
    ws = workspace(("bn", c), lib.odehip_bn_workspace_bytes(c) + 8 * c, dev)
    _lib.check(lib.odehip_bn_relu_up2x_backward(_ptr(grad_out), _ptr(x), n, c, h, w, _ptr(stats[0]), _ptr(stats[1]), _ptr(stats[2]), _ptr(stats[3]),
                                                int(training), int(bool(upsample)), _ptr(gx), _ptr(gw), _ptr(gb), _ptr(g_pre), _ptr(ws), ws.numel(),
                                                _stream()))
    return gx, gw, gb


# ---- the conv encoder / decoder either side of the path (models/ODEConvGRU.py:101-140), one fused launch each ------------------
_codec_packs = {}   # id(module) -> (weakref to module, stamp, pack tensor)


def _codec_layers(seq, conv_type, kernel, n_mid):
    """The (conv, act, conv[, act]) structure of the reference's Encoder / Decoder with n_downs = n_ups = 2, or None."""
    mods = list(seq)
    if len(mods) not in (3, 4) or not isinstance(mods[0], conv_type) or not isinstance(mods[2], conv_type):
        return None
    if not all(isinstance(m, torch.nn.LeakyReLU) for m in mods[1::2]):
        return None
    c1, c2 = mods[0], mods[2]
    for c in (c1, c2):
        if c.kernel_size != (kernel, kernel) or c.stride != (2, 2) or c.padding != (1, 1) or c.bias is None or c.groups != 1 \
                or c.dilation != (1, 1) or (conv_type is torch.nn.ConvTranspose2d and c.output_padding != (0, 0)):
            return None
    if c1.out_channels != n_mid or c2.in_channels != n_mid or len({m.negative_slope for m in mods[1::2]}) != 1:
        return None
    return c1, c2, float(mods[1].negative_slope)


def frame_encoder_supported(seq):
    ls = _codec_layers(seq, torch.nn.Conv2d, 3, 16)
    return ls is not None and len(list(seq)) == 4 and 1 <= ls[0].in_channels <= 4 and ls[1].out_channels in (32, 64, 128)


def frame_decoder_supported(seq):
    ls = _codec_layers(seq, torch.nn.ConvTranspose2d, 4, 32)
    return ls is not None and len(list(seq)) == 3 and ls[0].in_channels in (32, 64, 128) and 1 <= ls[1].out_channels <= 4


def _codec_pack(seq, c1, c2, n_floats, pack_fn, in_ch, out_ch):
    import weakref
    params = (c1.weight, c1.bias, c2.weight, c2.bias)
    for p in params:
        require_device_tensor(p, "codec parameter")
    stamp = tuple((p.data_ptr(), p._version) for p in params)
    ent = _codec_packs.get(id(seq))
    if ent is not None and ent[0]() is seq and ent[1] == stamp:
        return ent[2]
    pack = torch.empty(int(n_floats), dtype=torch.float32, device=c1.weight.device)
    _lib.check(pack_fn(_ptr(c1.weight.detach().contiguous()), _ptr(c1.bias.detach().contiguous()), _ptr(c2.weight.detach().contiguous()),
                       _ptr(c2.bias.detach().contiguous()), in_ch, out_ch, _ptr(pack), _stream()))
    if len(_codec_packs) > 32:
        for k in [k for k, v in _codec_packs.items() if v[0]() is None]:
            del _codec_packs[k]
    _codec_packs[id(seq)] = (weakref.ref(seq), stamp, pack)
    return pack


def frame_encode(seq, frames):
    """`seq` = the reference Encoder's nn.Sequential (n_downs = 2).  frames (B,T,c,64,64) -> (T,B,out_ch,16,16) contiguous,
    TIME-FIRST: the layout ODEConvGRU.py:64-68 reaches through a permuted view."""
    require_device_tensor(frames, "frames")
    if not frame_encoder_supported(seq):
        raise ValueError("frame_encode: not the reference's Encoder structure (Conv2d 3/2/1 -> LeakyReLU -> Conv2d 3/2/1 -> LeakyReLU)")
    c1, c2, slope = _codec_layers(seq, torch.nn.Conv2d, 3, 16)
    if frames.dim() != 5 or frames.shape[2] != c1.in_channels or tuple(frames.shape[3:]) != (64, 64):
        raise ValueError(f"frame_encode: frames must be (B,T,{c1.in_channels},64,64), got {tuple(frames.shape)}")
    frames = frames.detach().contiguous()
    b, t = frames.shape[:2]
    lib = _lib.load()
    pack = _codec_pack(seq, c1, c2, lib.odehip_frame_encoder_pack_floats(c1.in_channels, c2.out_channels), lib.odehip_pack_frame_encoder,
                       c1.in_channels, c2.out_channels)
    out = torch.empty((t, b, c2.out_channels, 16, 16), dtype=torch.float32, device=frames.device)
    _lib.check(lib.odehip_frame_encode(_ptr(pack), _ptr(frames), b, t, c1.in_channels, c2.out_channels, slope, _ptr(out), _stream()))
    return out


def frame_decode(seq, latents, apply_sigmoid, save_mid=False):
    """`seq` = the reference Decoder's nn.Sequential (n_ups = 2).  latents (..., C, 16, 16) (any leading dims, e.g. the solver's
    (T,B)) -> (..., out_ch, 64, 64); apply_sigmoid folds the F.sigmoid of ODEConvGRU.py:85 into the launch.  save_mid: also return
    the 32-channel intermediate (N, 32 x 32 x 32 floats, the layout of odehip_frame_decode_train) for the backward pass."""
    require_device_tensor(latents, "latents")
    if not frame_decoder_supported(seq):
        raise ValueError("frame_decode: not the reference's Decoder structure (ConvTranspose2d 4/2/1 -> LeakyReLU -> ConvTranspose2d 4/2/1)")
    c1, c2, slope = _codec_layers(seq, torch.nn.ConvTranspose2d, 4, 32)
    if latents.dim() < 4 or latents.shape[-3] != c1.in_channels or tuple(latents.shape[-2:]) != (16, 16):
        raise ValueError(f"frame_decode: latents must be (...,{c1.in_channels},16,16), got {tuple(latents.shape)}")
    latents = latents.detach().contiguous()
    lead = tuple(latents.shape[:-3])
    n = 1
    for d in lead:
        n *= d
    lib = _lib.load()
    pack = _codec_pack(seq, c1, c2, lib.odehip_frame_decoder_pack_floats(c1.in_channels, c2.out_channels), lib.odehip_pack_frame_decoder,
                       c1.in_channels, c2.out_channels)
    out = torch.empty(lead + (c2.out_channels, 64, 64), dtype=torch.float32, device=latents.device)
    mid = torch.empty((n, 32 * 32 * 32), dtype=torch.float32, device=latents.device) if save_mid else None
    _lib.check(lib.odehip_frame_decode_train(_ptr(pack), _ptr(latents), n, c1.in_channels, c2.out_channels, slope, 1 if apply_sigmoid else 0,
                                             _ptr(out), _ptr(mid) if save_mid else None, _stream()))
    return (out, mid) if save_mid else out



def codec_backward_enabled():
    """ODEHIP_CODEC_BACKWARD=0: the frame encoder / decoder run as library calls under autograd (the behaviour before round 3)."""
    return os.environ.get("ODEHIP_CODEC_BACKWARD", "1") != "0"


def frame_encoder_backward_supported(seq):
    """The shapes csrc/frame_codec_backward.hip implements: one frame channel, 32 / 64 latent channels."""
    if not frame_encoder_supported(seq):
        return False
    c1, c2, _ = _codec_layers(seq, torch.nn.Conv2d, 3, 16)
    return c1.in_channels == 1 and c2.out_channels in (32, 64)


def frame_decoder_backward_supported(seq):
    if not frame_decoder_supported(seq):
        return False
    c1, c2, _ = _codec_layers(seq, torch.nn.ConvTranspose2d, 4, 32)
    return c2.out_channels == 1 and c1.in_channels in (32, 64)


class _FrameEncodeFn(torch.autograd.Function):
    """frame_encode under autograd: forward = the fused launch, backward = odehip_frame_encode_backward (gradients of the four
    parameter tensors; the frames carry none -- the caller checks that they do not ask for one)."""

    @staticmethod
    def forward(ctx, seq, frames, w1, b1, w2, b2):
        out = frame_encode(seq, frames)
        ctx.seq = seq
        ctx.params = (w1, b1, w2, b2)   # the backward re-packs from the live parameters: they must still be the forward's
        ctx.versions = tuple(p._version for p in ctx.params)
        ctx.save_for_backward(frames, out, w2)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        if tuple(p._version for p in ctx.params) != ctx.versions:
            raise RuntimeError("a parameter of the frame encoder was modified in place between forward and backward")
        frames, out, w2 = ctx.saved_tensors
        seq = ctx.seq
        c1, c2, slope = _codec_layers(seq, torch.nn.Conv2d, 3, 16)
        lib = _lib.load()
        b, t = frames.shape[:2]
        pack = _codec_pack(seq, c1, c2, lib.odehip_frame_encoder_pack_floats(1, c2.out_channels), lib.odehip_pack_frame_encoder, 1, c2.out_channels)
        g = g.contiguous()
        frames = frames.detach().contiguous()
        dev = frames.device
        dw1, db1 = torch.empty_like(c1.weight), torch.empty_like(c1.bias)
        dw2, db2 = torch.empty_like(c2.weight), torch.empty_like(c2.bias)
        nws = int(lib.odehip_frame_encode_backward_workspace_floats(b, t, 1, c2.out_channels))
        ws = torch.empty(nws, dtype=torch.float32, device=dev)
        _lib.check(lib.odehip_frame_encode_backward(_ptr(pack), _ptr(w2.detach().contiguous()), _ptr(frames), _ptr(out), _ptr(g), b, t, 1,
                                                    c2.out_channels, slope, _ptr(dw1), _ptr(db1), _ptr(dw2), _ptr(db2), _ptr(ws), nws, _stream()))
        return None, None, dw1, db1, dw2, db2


class _FrameDecodeFn(torch.autograd.Function):
    """frame_decode (+ sigmoid) under autograd: backward = odehip_frame_decode_backward (latents' and the four parameters' gradients)."""

    @staticmethod
    def forward(ctx, seq, latents, apply_sigmoid, w1, b1, w2, b2):
        # ODEHIP_CODEC_SAVE_MID=0: keep nothing but inputs and outputs, the backward recomputes the intermediate (32 KiB per image less
        # memory, ~0.1 ms more per 640 images)
        if os.environ.get("ODEHIP_CODEC_SAVE_MID", "1") != "0":
            pred, mid = frame_decode(seq, latents, apply_sigmoid, save_mid=True)
        else:
            pred, mid = frame_decode(seq, latents, apply_sigmoid), None
        ctx.seq, ctx.apply_sigmoid, ctx.has_mid = seq, bool(apply_sigmoid), mid is not None
        ctx.params = (w1, b1, w2, b2)
        ctx.versions = tuple(p._version for p in ctx.params)
        ctx.save_for_backward(latents, pred, w1, *([mid] if mid is not None else []))
        return pred

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        if tuple(p._version for p in ctx.params) != ctx.versions:
            raise RuntimeError("a parameter of the frame decoder was modified in place between forward and backward")
        latents, pred, w1 = ctx.saved_tensors[:3]
        mid = ctx.saved_tensors[3] if ctx.has_mid else None
        seq = ctx.seq
        c1, c2, slope = _codec_layers(seq, torch.nn.ConvTranspose2d, 4, 32)
        lib = _lib.load()
        lat = latents.detach().contiguous()
        n = lat.numel() // (c1.in_channels * 256)
        pack = _codec_pack(seq, c1, c2, lib.odehip_frame_decoder_pack_floats(c1.in_channels, 1), lib.odehip_pack_frame_decoder, c1.in_channels, 1)
        g = g.contiguous()
        g_lat = torch.empty_like(lat)
        dw1, db1 = torch.empty_like(c1.weight), torch.empty_like(c1.bias)
        dw2, db2 = torch.empty_like(c2.weight), torch.empty_like(c2.bias)
        nws = int(lib.odehip_frame_decode_backward_workspace_floats(n, c1.in_channels, 1))
        ws = torch.empty(nws, dtype=torch.float32, device=lat.device)
        _lib.check(lib.odehip_frame_decode_backward(_ptr(pack), _ptr(w1.detach().contiguous()), _ptr(lat), _ptr(mid) if mid is not None else None,
                                                    _ptr(pred), _ptr(g), n, c1.in_channels, 1,
                                                    slope, 1 if ctx.apply_sigmoid else 0, _ptr(g_lat), _ptr(dw1), _ptr(db1), _ptr(dw2), _ptr(db2),
                                                    _ptr(ws), nws, _stream()))
        return None, g_lat.view_as(latents), None, dw1, db1, dw2, db2


def frame_encode_autograd(seq, frames):
    """frame_encode as a differentiable op of the encoder's parameters (frames must not require a gradient)."""
    if not frame_encoder_backward_supported(seq):
        raise ValueError("frame_encode_autograd: one frame channel and 32 / 64 latent channels only")
    if frames.requires_grad:
        raise ValueError("frame_encode_autograd: no gradient with respect to the frames (use the library path)")
    c1, c2, _ = _codec_layers(seq, torch.nn.Conv2d, 3, 16)
    return _FrameEncodeFn.apply(seq, frames, c1.weight, c1.bias, c2.weight, c2.bias)


def frame_decode_autograd(seq, latents, apply_sigmoid):
    """frame_decode as a differentiable op of the latents and the decoder's parameters."""
    if not frame_decoder_backward_supported(seq):
        raise ValueError("frame_decode_autograd: one frame channel and 32 / 64 latent channels only")
    c1, c2, _ = _codec_layers(seq, torch.nn.ConvTranspose2d, 4, 32)
    return _FrameDecodeFn.apply(seq, latents, apply_sigmoid, c1.weight, c1.bias, c2.weight, c2.bias)
