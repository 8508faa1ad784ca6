import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def _gpu_ready():
    import torch
    from pde_surrogate_amd import _lib
    return torch.cuda.is_available() and os.path.exists(_lib.LIB_PATH)


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests SKIP (not fail) on a machine without an MI355X or without a built libpdes_hip.so"""
    if _gpu_ready():
        return
    skip = pytest.mark.skip(reason='needs an MI355X and pde_surrogate_amd/libpdes_hip.so '
                                   '(python -m pde_surrogate_amd.build)')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def option():
    """set a kernel-selection option of the library (pdes_context_set_option) for one test; restored afterwards"""
    from pde_surrogate_amd import _lib
    touched = []

    def setter(key, value):
        touched.append(key)
        _lib.set_option(key, value)
    yield setter
    for k in touched:
        _lib.set_option(k, None)
        _lib._overrides.pop(k, None)
