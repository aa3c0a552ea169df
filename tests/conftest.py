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
