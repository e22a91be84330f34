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


def load_cases(npz_name):
    """Split an npz with keys 'case__field' into {case: {field: tensor}}."""
    z = np.load(os.path.join(GOLDEN, npz_name))
    cases = {}
    for k in z.files:
        c, f = k.split("__", 1)
        cases.setdefault(c, {})[f] = torch.from_numpy(z[k])
    return cases


def load_flat(npz_name):
    z = np.load(os.path.join(GOLDEN, npz_name))
    return {k: torch.from_numpy(z[k]) for k in z.files}


@pytest.fixture(scope="session")
def golden_attention():
    return load_cases("g3_bidaf_attention.npz")


@pytest.fixture(scope="session")
def golden_rnn():
    return load_cases("g4_rnn_encoder.npz")


@pytest.fixture(scope="session")
def golden_hot():
    return load_flat("g5_hot_region.npz")


def maxdiff(a, b):
    return (a.double() - b.double()).abs().max().item()


def scaled_tol(ref, tol=1e-4):
    """north_star tolerance: 1e-4 fp32, scaled by the tensor's magnitude when that exceeds 1
    (sums over B*T terms such as weight gradients grow with the problem size)."""
    return tol * max(1.0, ref.abs().max().item())
