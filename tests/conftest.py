import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


def state_dict_of(gold, prefix="sd."):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in gold.items() if k.startswith(prefix)}


def rel_l2(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def record(name, value):
    """Observed parity errors are appended to gpurun_out/parity_observed.jsonl (scratch; the per-round summary that the
    tolerances are calibrated against is committed under profiles/)."""
    import json
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_observed.jsonl"), "a") as fh:
            fh.write(json.dumps({"test": os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0], "name": name, "value": float(value)}) + "\n")
    except OSError:
        pass
    return value


def classic_rk4_step(func, t0, dt, t1, y0):
    """The classic Runge-Kutta 4 step (NOT what torchdiffeq's 'rk4' runs: that is the 3/8 rule).  Test-only: it exists so
    the suite can show that its fixtures tell the two apart."""
    k1 = func(t0, y0)
    k2 = func(t0 + dt / 2, y0 + dt * k1 / 2)
    k3 = func(t0 + dt / 2, y0 + dt * k2 / 2)
    k4 = func(t1, y0 + dt * k3)
    return (k1 + 2 * k2 + 2 * k3 + k4) * dt / 6, k1


def vigorous_case():
    """Inputs of tests/golden/traj_vig.npz: weights of f_A.npz times `scale`, z0 of traj_A.npz, 4 steps of 0.2."""
    fa, tr, vg = load_golden("f_A.npz"), load_golden("traj_A.npz"), load_golden("traj_vig.npz")
    sd = {k: v * float(vg["scale"]) for k, v in state_dict_of(fa).items()}
    return sd, torch.from_numpy(tr["z0"]), torch.from_numpy(vg["t"]), vg


@pytest.fixture(scope="session")
def cuda():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _workspace_guards(request):
    """After every GPU test: no HIP kernel may have written past the workspace it was given (hip_ops.check_canaries)."""
    yield
    if request.node.get_closest_marker("gpu") is not None and torch.cuda.is_available():
        from ode_rl_amd import hip_ops
        torch.cuda.synchronize()
        hip_ops.check_canaries()
