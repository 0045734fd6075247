"""pytest configuration: markers, import paths, shared fixtures.

`-m "not gpu"`: oracle vs golden vectors / independent restatement, host logic, C-ABI load + exports.
`-m gpu`      : parity of the HIP engine (through the C ABI) against the oracle.
Nothing here reads /root/reference: that tree does not exist on the GPU box.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "ocean-perception_amd", "python"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    config.addinivalue_line("markers", "rate: asserts on a measured rate (wall clock); collected LAST so that a noisy box "
                                       "cannot cost a -x run the parity tests behind it")


def pytest_collection_modifyitems(config, items):
    items.sort(key=lambda it: 1 if it.get_closest_marker("rate") else 0)  # stable: everything else keeps its order


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.load()
    return oracle_lib


@pytest.fixture(scope="session")
def synth():
    import synth as s
    return s


@pytest.fixture(scope="session")
def pm():
    import pm_ctypes
    pm_ctypes.load()
    return pm_ctypes


def small_pair(synth_mod, index, rows, cols, **kw):
    p = synth_mod.make_pair(index, rows=rows, cols=cols, **kw)
    return p["left"], p["right"], p["seed_l"], p["seed_r"], p["gt"]


def assert_same(a, b, what=""):
    """Bit-for-bit equality of float32 maps (values; +0 == -0 is not expected to occur)."""
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    if not np.array_equal(a, b):
        bad = np.argwhere(a != b)
        y, x = bad[0][:2]
        raise AssertionError(f"{what}: {len(bad)} of {a.size} values differ; first at (y={y}, x={x}): "
                             f"{a[y, x]!r} vs {b[y, x]!r}; max |diff| {np.abs(a.astype(np.float64) - b).max()}")
