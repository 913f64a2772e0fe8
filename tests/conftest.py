import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, e.g. a plain `pytest tests/`."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _split_kernels_at_test_sizes():
    """The split-arithmetic kernels are only SELECTED for launches that fill the chip (>= 192 one-per-CU workgroups,
    lf_tapgemm_split_ok); the small-shape parity tests lift that rule through the debug hook (csrc/lf_debug.h).  The
    full-size tests (test_baseline_configs_gpu.py) switch it back off and run the shipped selection."""
    import torch
    if torch.cuda.is_available():
        from lanedetection_end2end_amd import _lib
        _lib.load().lf_debug_set_split_any_size(1)
    yield


@pytest.fixture(scope="session")
def golden_fit():
    return np.load(os.path.join(GOLDEN, "fit_head.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden_backbone():
    return np.load(os.path.join(GOLDEN, "backbone.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden_e2e():
    return np.load(os.path.join(GOLDEN, "e2e.npz"), allow_pickle=False)


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def tie_tolerant_err(a, b, tol):
    """(fraction of elements further than tol * max|b| from b, relative L2 error).  For gradients that pass through ReLUs of
    ~1e6 pre-activations: the smallest |pre-activation| of such a tensor is ~1e-6, inside fp32 rounding, so an fp32 forward may
    take the other branch of ONE ReLU than the fp64 oracle; the gradient then differs on the receptive field of that element
    (a 5x5 patch per 3x3 layer below it) and nowhere else.  A wrong kernel fails both numbers; a flipped tie moves neither."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = np.abs(a - b)
    return float((d > tol * np.abs(b).max()).mean()), float(np.sqrt((d ** 2).sum() / max((b ** 2).sum(), 1e-300)))
