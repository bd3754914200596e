import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


@pytest.fixture(scope="session")
def golden_seed():
    with open(os.path.join(GOLDEN, "meta.json")) as f:
        return json.load(f)["weights_seed"]


@pytest.fixture(scope="session")
def dit_weights_np(golden_seed):
    from smalltts_amd.weights import dit_param_specs, synth_state_dict
    return synth_state_dict(dit_param_specs(), golden_seed)


@pytest.fixture(scope="session")
def dit_weights(dit_weights_np):
    from oracle.dit_oracle import to_torch
    return to_torch(dit_weights_np)


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
