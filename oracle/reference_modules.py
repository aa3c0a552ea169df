"""ORACLE (test infrastructure only) -- CPU restatement of the reference-authored modules.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this.
Pinned by fixtures generated from the reference's own classes (tests/golden/make_golden.py
imports /root/reference with two import stubs; fixtures F1-F4, F7 of SURVEY.md section 8c).

Everything is functional: weights come in as plain lists / dicts of tensors so the same
function checks both the HIP path and the reference's `state_dict` layout.
"""
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- dynamics f
class _MixedConv(torch.autograd.Function):
    """nn.Conv2d with bf16 operands and fp32 accumulation, as the HIP bf16 path computes it (BASELINE.json configs[4]):
    forward conv(bf16(x), bf16(W)) + b in fp32; backward: input gradient from bf16(g) and bf16(W); weight gradient from
    bf16(x) and bf16(g) (fp32 accumulation); bias gradient from the unrounded g."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return F.conv2d(x.bfloat16().float(), w.bfloat16().float(), b, stride=1, padding=w.shape[-1] // 2)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        pad = w.shape[-1] // 2
        gx = torch.nn.grad.conv2d_input(x.shape, w.bfloat16().float(), g.bfloat16().float(), padding=pad)
        gw = torch.nn.grad.conv2d_weight(x.bfloat16().float(), w.shape, g.bfloat16().float(), padding=pad)
        return gx, gw, g.sum((0, 2, 3))


def convnet_forward(y, weights, biases, final_tanh=False, compute_dtype="f32"):
    """`helpers/utils.py:158-183` create_convnet with nonlinear='relu':
    Conv3x3 -> [ReLU -> Conv3x3]*n_layers -> ReLU -> Conv3x3 [-> Tanh].

    `weights[i]` is (Cout, Cin, k, k) with 'same' padding, stride 1.
    """
    x = y
    n = len(weights)
    for i, (w, b) in enumerate(zip(weights, biases)):
        if compute_dtype == "bf16":
            x = _MixedConv.apply(x, w, b)
        else:
            x = F.conv2d(x, w, b, stride=1, padding=w.shape[-1] // 2)
        if i < n - 1:
            x = torch.relu(x)
    if final_tanh:
        x = torch.tanh(x)
    return x


def ode_func(weights, biases, backwards=False, final_tanh=False, compute_dtype="f32"):
    """`modules/DiffEqSolver.py:57-80` ODEFunc.forward as a closure f(t, y); t is ignored."""
    def f(t, y):
        g = convnet_forward(y, weights, biases, final_tanh, compute_dtype)
        return -g if backwards else g
    return f


def split_convnet_state(sd, prefix=""):
    """Conv layers of a create_convnet Sequential sit at indices 0,2,4,... (`helpers/utils.py:166-177`)."""
    ws, bs, i = [], [], 0
    while f"{prefix}{i}.weight" in sd:
        ws.append(sd[f"{prefix}{i}.weight"])
        bs.append(sd[f"{prefix}{i}.bias"])
        i += 2
    return ws, bs


# --------------------------------------------------------------------------- ConvGRU cell
def convgru_cell(x, h, p, compute_dtype="f32"):
    """One step of `modules/ConvGRUCell.py:72-82`.

    p: dict with conv_gates.0.{weight,bias}, conv_gates.1.{weight,bias} (GroupNorm affine),
    conv_can.0.*, conv_can.1.*.  GroupNorm groups are 2*hid//32 and hid//32 (`:40-50`),
    eps 1e-5.  Gate order: z first, r second (`:75-77`).
    """
    hid = h.shape[1]
    pad = p["conv_gates.0.weight"].shape[-1] // 2
    def conv(inp, w, b):   # compute_dtype="bf16": the HIP bf16 path's arithmetic (bf16 operands, fp32 accumulation)
        return _MixedConv.apply(inp, w, b) if compute_dtype == "bf16" else F.conv2d(inp, w, b, padding=pad)
    g = conv(torch.cat((x, h), 1), p["conv_gates.0.weight"], p["conv_gates.0.bias"])
    g = F.group_norm(g, 2 * hid // 32, p["conv_gates.1.weight"], p["conv_gates.1.bias"], eps=1e-5)
    zg, rg = torch.split(g, hid, dim=1)
    z, r = torch.sigmoid(zg), torch.sigmoid(rg)
    c = conv(torch.cat((x, r * h), 1), p["conv_can.0.weight"], p["conv_can.0.bias"])
    c = torch.tanh(F.group_norm(c, hid // 32, p["conv_can.1.weight"], p["conv_can.1.bias"], eps=1e-5))
    return (1 - z) * h + z * c


# --------------------------------------------------------------------------- encoder loop
def ode_convgru_encode(inputs, timesteps, f_enc, cell_params, head_params, compute_dtype="f32", run_backwards=True):
    """`modules/ODEConvGRUCell.py:32-78`: reverse-time explicit-Euler + ConvGRU loop, then the
    1x1 -> ReLU -> 1x1 head, split into (mean_z0, |std_z0|).

    inputs (T,B,C,H,W) time-first; timesteps (T,) float64.  Returns mean, std, latent_ys(B,T,...).
    """
    T, B, C, H, W = inputs.shape
    assert T == len(timesteps), "Sequence length should be same as time_steps"
    prev = torch.zeros((B, C, H, W), dtype=inputs.dtype)
    prev_t, t_i = timesteps[-1] + 0.01, timesteps[-1]
    ys = []
    for i in (reversed(range(T)) if run_backwards else range(T)):   # :49-51
        inc = f_enc(prev_t, prev) * (t_i - prev_t)
        assert not torch.isnan(inc).any()
        ode_sol = prev + inc
        yi = convgru_cell(inputs[i], ode_sol, cell_params, compute_dtype)
        prev = yi
        prev_t, t_i = timesteps[i], timesteps[i - 1]
        ys.append(yi)
    latent = torch.stack(ys, 0).permute(1, 0, 2, 3, 4)
    z = F.conv2d(yi, head_params["0.weight"], head_params["0.bias"])
    z = F.conv2d(torch.relu(z), head_params["2.weight"], head_params["2.bias"])
    out_ch = z.shape[1] // 2
    mean, std = torch.split(z, out_ch, dim=1)
    return mean, std.abs(), latent
