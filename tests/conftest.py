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


def procedural_tensor(shape, seed, lo, hi):
    """A float32 tensor defined by its indices alone (splitmix64 of element index and seed, top 24 bits -> uniform in [lo, hi)):
    pure integer arithmetic, so make_golden.py here and the tests on the GPU box build bit-identical weights and inputs and
    the fixtures only need to store OUTPUTS."""
    n = 1
    for d in shape:
        n *= int(d)
    with np.errstate(over="ignore"):
        x = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed) * np.uint64(0xD1B54A32D192ED03) + np.uint64(0x2545F4914F6CDD1D)
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    u = (x >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32).reshape(tuple(shape)))


def procedural_state_dict(reference_sd, seed):
    """Procedural values for every entry of a state_dict (shapes and keys taken from `reference_sd`): conv weights / biases
    uniform within +-1/sqrt(fan_in) (PyTorch's default range), normalisation weights in [0.7, 1.3], their biases and running
    means in [-0.3, 0.3], running variances in [0.5, 1.5]; integer buffers (num_batches_tracked) are left as they are."""
    import zlib
    out = {}
    for k, v in reference_sd.items():
        if not torch.is_floating_point(v):
            out[k] = v.clone()
            continue
        s = (zlib.crc32(k.encode()) + 7919 * seed) & 0x7FFFFFFF
        if v.dim() >= 2:
            b = 1.0 / float(np.sqrt(np.prod(v.shape[1:])))
            out[k] = procedural_tensor(v.shape, s, -b, b)
        elif k.endswith("running_var"):
            out[k] = procedural_tensor(v.shape, s, 0.5, 1.5)
        elif k.endswith("running_mean"):
            out[k] = procedural_tensor(v.shape, s, -0.3, 0.3)
        elif _is_norm_weight(k, reference_sd):
            out[k] = procedural_tensor(v.shape, s, 0.7, 1.3)
        elif k.endswith("weight"):      # cannot happen (1-D weights belong to normalisation layers)
            out[k] = procedural_tensor(v.shape, s, 0.7, 1.3)
        else:   # biases: conv biases within the default range of a 3x3 64-channel layer, normalisation biases alike
            out[k] = procedural_tensor(v.shape, s, -0.3, 0.3) if _is_norm_bias(k, reference_sd) else procedural_tensor(v.shape, s, -0.04, 0.04)
    return out


VIDODE_FLOW_GAIN = 10.0   # fixture F11: the decoder's last conv is scaled so that flows reach several pixels (warp + border clamp exercised)


def vidode_state_dict(reference_sd, seed=14):
    sd = procedural_state_dict(reference_sd, seed)
    for k in ("conv_decoder.cnn_decoder.8.weight", "conv_decoder.cnn_decoder.8.bias"):
        sd[k] = sd[k] * VIDODE_FLOW_GAIN
    return sd


def _is_norm_weight(k, sd):
    return k.endswith(".weight") and sd[k].dim() == 1


def _is_norm_bias(k, sd):
    w = k[:-len("bias")] + "weight"
    return k.endswith(".bias") and w in sd and sd[w].dim() == 1


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
