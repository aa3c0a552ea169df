"""Generate the golden fixtures in this directory (dev container only; never runs on the GPU box).

Imports the REFERENCE's own classes from /root/reference (read-only, never copied) with two
import stubs for packages the image lacks: `skimage` (only used by an
SSIM metric off the hot path) and `torchdiffeq` (un-vendored; `odeint` is routed to the
restatement in oracle/torchdiffeq_ref.py, which is why fixtures F5/F6 pin the *wiring* of
`DiffEqSolver` and not torchdiffeq's internals -- see the "parity unpinned" note there).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
Writes .npz files (fp32 inputs + expected outputs) next to this script.
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import torchdiffeq_ref  # noqa: E402


VIG_SCALE = 2.5
VIG_T = [0.0, 0.2, 0.4, 0.6, 0.8]


def _install_stubs():
    sk, skm = types.ModuleType("skimage"), types.ModuleType("skimage.metrics")
    skm.structural_similarity = None
    sk.metrics = skm
    sys.modules["skimage"], sys.modules["skimage.metrics"] = sk, skm
    td = types.ModuleType("torchdiffeq")
    td.odeint = torchdiffeq_ref.odeint
    sys.modules["torchdiffeq"] = td
    sys.path.insert(0, "/root/reference")


def _np(sd, prefix=""):
    return {prefix + k: v.detach().cpu().numpy() for k, v in sd.items()}


def main():
    _install_stubs()
    from modules.DiffEqSolver import ODEFunc, DiffEqSolver  # reference
    from modules.ConvGRUCell import ConvGRUCell  # reference
    from modules.ODEConvGRUCell import ODEConvGRUCell  # reference
    from models.ODEConvGRU import ODEConvGRU  # reference

    torch.set_grad_enabled(False)
    dev = torch.device("cpu")

    # F1: dynamics f, ODEConvGRU shape (5x conv3x3 64->64)
    torch.manual_seed(0)
    fA = ODEFunc(n_inputs=64, n_outputs=64, n_layers=3, n_units=64, downsize=False,
                 nonlinear="relu", final_act=False, device=dev)
    y = torch.randn(2, 64, 16, 16) * 0.5
    np.savez(os.path.join(HERE, "f_A.npz"), y=y.numpy(), out=fA(0.0, y).numpy(),
             out_backwards=fA(0.0, y, backwards=True).numpy(), **_np(fA.state_dict(), "sd."))

    # F2: dynamics f, VidODE shape (128->64->64->64->128)
    torch.manual_seed(1)
    fV = ODEFunc(n_inputs=128, n_outputs=128, n_layers=2, n_units=64, downsize=False,
                 nonlinear="relu", final_act=False, device=dev)
    yv = torch.randn(1, 128, 16, 16) * 0.5
    np.savez(os.path.join(HERE, "f_V.npz"), y=yv.numpy(), out=fV(0.0, yv).numpy(),
             **_np(fV.state_dict(), "sd."))

    # F3: ConvGRU cell (reduced: 32 channels) with non-trivial GroupNorm affine
    torch.manual_seed(2)
    cell = ConvGRUCell((16, 16), 32, 32, 5)
    for k, p in cell.state_dict().items():
        if k.endswith("1.weight") or k.endswith("1.bias"):
            p.copy_(torch.randn_like(p) * 0.3 + (1.0 if k.endswith("weight") else 0.0))
    x, h = torch.randn(2, 32, 16, 16) * 0.5, torch.randn(2, 32, 16, 16) * 0.5
    _, h1 = cell(input_tensor=x[None], h_cur=h, seq_len=1)
    np.savez(os.path.join(HERE, "cgru.npz"), x=x.numpy(), h=h.numpy(), out=h1.numpy(),
             **_np(cell.state_dict(), "sd."))

    # F4: ODEConvGRUCell encoder loop (reduced: 32 channels, 4 frames)
    torch.manual_seed(3)
    fE = ODEFunc(n_inputs=32, n_outputs=32, n_layers=3, n_units=32, downsize=False,
                 nonlinear="relu", final_act=False, device=dev)
    enc = ODEConvGRUCell(fE, None, (16, 16), 32, device=dev)
    for k, p in enc.state_dict().items():
        if "cgru_cell" in k and (k.endswith("1.weight") or k.endswith("1.bias")):
            p.copy_(torch.randn_like(p) * 0.3 + (1.0 if k.endswith("weight") else 0.0))
    inp = torch.randn(4, 2, 32, 16, 16) * 0.5
    tt = torch.tensor(np.arange(4) / 8)
    mean, std = enc(inp, tt)
    _, latent = enc.run_ode_conv_gru(inp, tt)
    np.savez(os.path.join(HERE, "encode.npz"), inputs=inp.numpy(), t=tt.numpy(), mean=mean.numpy(),
             std=std.numpy(), latent=latent.numpy(), **_np(enc.state_dict(), "sd."))

    # F5/F6: reference DiffEqSolver wiring (odeint := restatement), weights of F1
    z0 = torch.randn(2, 64, 16, 16, generator=torch.Generator().manual_seed(1234)) * 0.5
    t = torch.tensor(np.arange(10, 20) / 20)
    out = {"z0": z0.numpy(), "t": t.numpy()}
    for method in ("rk4", "euler", "midpoint", "dopri5"):
        solver = DiffEqSolver(fA, method, device=dev)
        sol = solver(z0, t)
        assert sol.shape == (10, 2, 64, 16, 16)
        out[f"{method}.first"] = sol[1].numpy()
        out[f"{method}.last"] = sol[-1].numpy()
        out[f"{method}.norms"] = sol.flatten(1).norm(dim=1).numpy()
    st = {}
    sol = torchdiffeq_ref.odeint(fA, z0, t, rtol=1e-4, atol=1e-5, method="dopri5", stats=st)
    out["dopri5.nfe"] = np.array([st["nfe"], st.get("n_accept", 0), st.get("n_reject", 0)])
    out["dopri5.dts"] = np.array(st["dts"])
    solver_mem = DiffEqSolver(fA, "rk4", device=dev, memory=True)
    mem = solver_mem(z0, t[:3])
    out["memory.shape"] = np.array(mem.shape)
    out["memory.last"] = mem[:, -1].numpy()
    np.savez(os.path.join(HERE, "traj_A.npz"), **out)

    # F5v: the same wiring on VIGOROUS dynamics.  On traj_A the state moves 2.4 % and euler / midpoint / rk4 agree to 1e-5,
    # so a wrong tableau passes there.  Here the weights of F1 are scaled by VIG_SCALE and the grid is 4 steps of 0.2: the
    # state moves ~140 % with half the ReLUs active, and euler / midpoint / classic RK4 / the 3/8 rule differ pairwise by
    # >= 1e-3 (asserted in tests/test_oracle_golden.py), while a relative perturbation of z0 is not amplified (so fp32
    # implementations still agree to ~1e-6).  Only outputs are stored: weights = f_A.npz * scale, z0 = traj_A.npz.
    keep = [p_.detach().clone() for p_ in fA.parameters()]
    with torch.no_grad():
        for p_ in fA.parameters():
            p_.mul_(VIG_SCALE)
    tv = torch.tensor(VIG_T, dtype=torch.float64)
    out = {"scale": np.float64(VIG_SCALE), "t": tv.numpy()}
    for method in ("rk4", "euler", "midpoint", "dopri5"):
        sol = DiffEqSolver(fA, method, device=dev)(z0, tv)
        out[f"{method}.first"] = sol[1].numpy()
        out[f"{method}.last"] = sol[-1].numpy()
        out[f"{method}.norms"] = sol.flatten(1).norm(dim=1).numpy()
    st = {}
    torchdiffeq_ref.odeint(fA, z0, tv, rtol=1e-4, atol=1e-5, method="dopri5", stats=st)
    out["dopri5.nfe"] = np.array([st["nfe"], st.get("n_accept", 0), st.get("n_reject", 0)])
    np.savez(os.path.join(HERE, "traj_vig.npz"), **out)
    with torch.no_grad():
        for p_, k_ in zip(fA.parameters(), keep):
            p_.copy_(k_)

    # F7: full ODEConvGRU.forward on a reduced config (32-channel latents), rk4
    torch.manual_seed(4)
    opt = argparse.Namespace(resolution=64, n_downs=2, conv_encoder_out_ch=32, in_channels=1,
                             n_ode_layers=3, neural_ode_n_units=32, neural_ode_decoder_out_ch=32,
                             decode_diff_method="rk4", mem=False, z_sample=False)
    model = ODEConvGRU(opt, dev)
    frames = torch.rand(2, 4, 1, 64, 64)
    ts = torch.tensor(np.arange(8) / 8)
    pred = model(frames, {"observed_tp": ts[:4], "tp_to_predict": ts[4:]})
    np.savez(os.path.join(HERE, "model.npz"), frames=frames.numpy(), t=ts.numpy(), pred=pred.numpy(),
             keys=np.array(sorted(model.state_dict().keys())), **_np(model.state_dict(), "sd."))

    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
