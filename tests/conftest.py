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


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def rel_err(a, b):
    """max|a-b| / max|b| -- the 'relative' metric of BASELINE.json's parity bar."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def rel_err_by_level(a, b, num_levels, axis):
    """rel_err per pyramid level: the channel axis (length num_levels * rd * rd, level-major) is split per level and
    every level is judged against ITS OWN max|b| -- level 3 is a mean of 64 cells, ~8x smaller than level 0, so
    the global metric alone would let an error of 8e-5 of level 3's scale pass a 1e-5 bar."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape and a.shape[axis] % num_levels == 0, (a.shape, b.shape, num_levels)
    return [rel_err(x, y) for x, y in zip(np.split(a, num_levels, axis=axis), np.split(b, num_levels, axis=axis))]


def rel_err_by_item(a, b):
    """rel_err per index of axis 0 (gradients: per batch item, each against its own max|b|)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    return [rel_err(a[i], b[i]) for i in range(a.shape[0])]


# measured errors of the GPU parity tests, written to gpurun_out/parity_errors.json at the end of the session (copied
# to profiles/ per round)
PARITY_LOG = {}


def log_errors(name, **errs):
    PARITY_LOG[name] = {k: ([float(x) for x in v] if isinstance(v, (list, tuple)) else float(v)) for k, v in errs.items()}


def pytest_sessionfinish(session, exitstatus):
    if not PARITY_LOG:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_errors.json"), "w") as f:
            json.dump(PARITY_LOG, f, indent=1, sort_keys=True)
    except OSError:
        pass


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def dev():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    from raft_amd import _lib
    _lib.load()                                   # fail loudly if the extension is missing
    return torch.device("cuda:0")
