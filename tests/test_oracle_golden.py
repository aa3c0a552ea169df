"""The oracle's restatement of the reference-authored modules against fixtures generated from the reference's OWN
classes (tests/golden/make_golden.py).  fp32 on CPU both sides; tolerance 1e-6 rel-L2 (same ATen kernels, op
order differs only in the Euler update and the functional GroupNorm call)."""
import numpy as np
import torch

from conftest import classic_rk4_step, load_golden, rel_l2, state_dict_of, vigorous_case
from oracle import reference_modules as rm
from oracle import torchdiffeq_ref as td


def test_f_A_and_f_V():
    for name in ("f_A.npz", "f_V.npz"):
        g = load_golden(name)
        ws, bs = rm.split_convnet_state(state_dict_of(g), "gradient_net.")
        y = torch.from_numpy(g["y"])
        assert rel_l2(rm.convnet_forward(y, ws, bs), torch.from_numpy(g["out"])) <= 1e-6
    g = load_golden("f_A.npz")
    ws, bs = rm.split_convnet_state(state_dict_of(g), "gradient_net.")
    assert len(ws) == 5 and all(w.shape == (64, 64, 3, 3) for w in ws)
    out_b = rm.ode_func(ws, bs, backwards=True)(0.0, torch.from_numpy(g["y"]))
    assert rel_l2(out_b, torch.from_numpy(g["out_backwards"])) <= 1e-6


def test_convgru_cell():
    g = load_golden("cgru.npz")
    out = rm.convgru_cell(torch.from_numpy(g["x"]), torch.from_numpy(g["h"]), state_dict_of(g))
    assert rel_l2(out, torch.from_numpy(g["out"])) <= 1e-6


def test_encoder_loop():
    g = load_golden("encode.npz")
    sd = state_dict_of(g)
    # aliased keys: the shared ODEFunc is registered under the cell as `ode_func`
    ws, bs = rm.split_convnet_state(sd, "ode_func.gradient_net.")
    cell = {k[len("cgru_cell."):]: v for k, v in sd.items() if k.startswith("cgru_cell.")}
    head = {k[len("transform_z0."):]: v for k, v in sd.items() if k.startswith("transform_z0.")}
    mean, std, latent = rm.ode_convgru_encode(torch.from_numpy(g["inputs"]), torch.from_numpy(g["t"]),
                                              rm.ode_func(ws, bs), cell, head)
    assert rel_l2(mean, torch.from_numpy(g["mean"])) <= 1e-6
    assert rel_l2(std, torch.from_numpy(g["std"])) <= 1e-6
    assert rel_l2(latent, torch.from_numpy(g["latent"])) <= 1e-6
    assert bool((std >= 0).all())


def test_solver_wiring_fixture():
    """F5/F6: the reference's DiffEqSolver returns time-first (T,B,C,H,W) with out[0] = z0; dopri5 stats."""
    fa, tr = load_golden("f_A.npz"), load_golden("traj_A.npz")
    ws, bs = rm.split_convnet_state(state_dict_of(fa), "gradient_net.")
    f = rm.ode_func(ws, bs)
    z0, t = torch.from_numpy(tr["z0"]), torch.from_numpy(tr["t"])
    assert t.dtype == torch.float64
    with torch.no_grad():
        for m in ("rk4", "euler", "midpoint"):
            sol = td.odeint(f, z0, t, rtol=1e-4, atol=1e-5, method=m)
            assert torch.equal(sol[0], z0)
            assert rel_l2(sol[-1], torch.from_numpy(tr[f"{m}.last"])) <= 1e-6
            np.testing.assert_allclose(sol.flatten(1).norm(dim=1).numpy(), tr[f"{m}.norms"], rtol=1e-5)
        st = {}
        sol = td.odeint(f, z0, t, rtol=1e-4, atol=1e-5, method="dopri5", stats=st)
    assert rel_l2(sol[-1], torch.from_numpy(tr["dopri5.last"])) <= 1e-5
    assert [st["nfe"], st.get("n_accept", 0), st.get("n_reject", 0)] == tr["dopri5.nfe"].tolist()
    # memory=True branch (reference :30-42): odeint on a 1-element t returns its input => h_next = 2 h_prev, batch-first
    assert tr["memory.shape"].tolist() == [2, 3, 64, 16, 16]
    assert rel_l2(torch.from_numpy(tr["memory.last"]), 8 * z0) <= 1e-7


def test_vigorous_fixture_discriminates_the_methods():
    """F5v (tests/golden/traj_vig.npz, generated through the reference's own DiffEqSolver): the state moves by more than its own
    norm, ReLUs are active, and euler / midpoint / classic RK4 / the 3/8 rule differ PAIRWISE by >= 1e-3 -- so the GPU test that
    asserts <= 1e-5 against this fixture fails for a wrong tableau, a wrong stage coefficient or the wrong method."""
    sd, z0, t, vg = vigorous_case()
    ws, bs = rm.split_convnet_state(sd, "gradient_net.")
    f = rm.ode_func(ws, bs)
    td._FIXED["_classic_rk4"] = classic_rk4_step
    try:
        with torch.no_grad():
            sols = {m: td._integrate_fixed(f, z0, t, m, {}) for m in ("euler", "midpoint", "rk4", "_classic_rk4")}
            st = {}
            sols["dopri5"] = td.odeint(f, z0, t, rtol=1e-4, atol=1e-5, method="dopri5", stats=st)
    finally:
        del td._FIXED["_classic_rk4"]
    for m in ("euler", "midpoint", "rk4", "dopri5"):   # the restatement reproduces the fixture
        assert rel_l2(sols[m][-1], torch.from_numpy(vg[f"{m}.last"])) <= 1e-6
        assert rel_l2(sols[m][1], torch.from_numpy(vg[f"{m}.first"])) <= 1e-6
    assert [st["nfe"], st.get("n_accept", 0), st.get("n_reject", 0)] == vg["dopri5.nfe"].tolist()
    move = float((sols["rk4"][-1] - z0).norm() / z0.norm())
    assert move >= 1.0, move
    hidden = torch.relu(torch.nn.functional.conv2d(sols["rk4"][-1], ws[0], bs[0], padding=1))
    assert 0.3 <= float((hidden > 0).float().mean()) <= 0.7
    names = ["euler", "midpoint", "rk4", "_classic_rk4"]
    for i, a in enumerate(names):
        for b in names[i + 1:]:
            assert rel_l2(sols[a][-1], sols[b][-1]) >= 1e-3, (a, b)
            assert rel_l2(sols[a][1], sols[b][1]) >= 1e-4, (a, b)   # already after ONE step
    assert rel_l2(sols["dopri5"][-1], sols["rk4"][-1]) >= 2e-4
