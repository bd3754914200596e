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
    # The CPU oracles are graphs of small torch ops: on the 256-core GPU box torch's default (all cores) is 3-30x SLOWER than 16
    # threads (bench.py's thread probe: 176 ms per denoiser call at 16 threads, 483 at 64, minutes at 256), and the oracle side is
    # most of the GPU suite's wall time.  SMTTS_TEST_THREADS overrides.
    try:
        import torch
        n = int(os.environ.get("SMTTS_TEST_THREADS", "0")) or min(16, os.cpu_count() or 1)
        torch.set_num_threads(max(1, n))
    except Exception:
        pass


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
