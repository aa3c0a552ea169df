"""Thin torch-facing wrappers over the C ABI: pointer plumbing, workspace and packed-weight caches.

Nothing here computes: every function hands raw device pointers to libodecgru_hip.so on torch's
current stream.  Tensors must be CUDA fp32; anything else raises (there is no CPU fallback).
"""
import ctypes

import torch

from . import _lib

_workspaces = {}


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def require_device_tensor(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} is on {t.device}: the HIP path needs CUDA (ROCm) tensors and has no CPU fallback")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32 (got {t.dtype})")


def workspace(key, nbytes, device):
    """Persistent per-shape scratch so that pointers stay stable across calls (graph replay)."""
    k = (key, device)
    buf = _workspaces.get(k)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1024), dtype=torch.uint8, device=device)
        _workspaces[k] = buf
    return buf


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


def conv_q4(src1, w_packed, bias, cout, ks, src2=None, relu=False):
    """One conv layer on Q4 activations (tests / building block)."""
    require_device_tensor(src1, "src1")
    b = src1.shape[0]
    cin1 = src1.shape[1] * 4
    cin = cin1 + (src2.shape[1] * 4 if src2 is not None else 0)
    dst = torch.empty((b, cout // 4, 256, 4), dtype=torch.float32, device=src1.device)
    d = _lib.ConvDesc(src1=src1.data_ptr(), src2=src2.data_ptr() if src2 is not None else None, cin1=cin1, cin=cin,
                      cout=cout, ks=ks, batch=b, w_packed=w_packed.data_ptr(),
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
        self._stamp = None
        self._packed = None
        self._bias = None
        self.desc = None

    def _current_stamp(self):
        return tuple((c.weight.data_ptr(), c.weight._version, c.bias.data_ptr(), c.bias._version) for c in self.convs)

    def refresh(self):
        stamp = self._current_stamp()
        if stamp == self._stamp:
            return self.desc
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
        self._packed = [pack_conv_weight(c.weight) for c in convs]
        self._bias = [c.bias.detach().contiguous() for c in convs]
        d = _lib.ConvStack()
        d.n_convs = len(convs)
        d.ks = ks
        d.channels[0] = convs[0].in_channels
        for i, c in enumerate(convs):
            d.channels[i + 1] = c.out_channels
            d.w_packed[i] = self._packed[i].data_ptr()
            d.bias[i] = self._bias[i].data_ptr()
        d.final_tanh = int(self.final_tanh)
        self.desc = d
        self._stamp = stamp
        return d


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


def odeint_fixed(stack, method, z0, t):
    """Whole fixed-grid trajectory in one C-ABI call.  Returns (T,B,C,16,16)."""
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
    nbytes = lib.odehip_odeint_workspace_bytes(ctypes.byref(desc), b, n, m, 0)
    ws = workspace(("odeint", b, n, m, tuple(desc.channels)), nbytes, z0.device)
    out = torch.empty((n, b, c, 16, 16), dtype=torch.float32, device=z0.device)
    tarr = (ctypes.c_double * n)(*t64)
    _lib.check(lib.odehip_odeint_fixed(ctypes.byref(desc), m, _ptr(z0), tarr, n, b, _ptr(out), 0, _ptr(ws), ws.numel(),
                                       _stream()))
    return out


def odeint_dopri5(stack, z0, t, rtol, atol, first_step=0.0, max_steps=0):
    """Adaptive dopri5 trajectory; returns ((T,B,C,16,16), stats dict)."""
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
    _lib.check(lib.odehip_odeint_dopri5(ctypes.byref(desc), _ptr(z0), tarr, n, b, float(rtol), float(atol), float(first_step or 0.0), int(max_steps),
                                        _ptr(out), stats, _ptr(ws), ws.numel(), _stream()))
    return out, {"nfe": stats[0], "n_accept": stats[1], "n_reject": stats[2], "attempts_enqueued": stats[3]}
