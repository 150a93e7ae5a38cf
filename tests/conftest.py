import json
import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401  -- before libmorna_hip.so is loaded: torch bundles its own libamdhip64, and a process that loads
#                              the system one first (through our library) and torch's afterwards ends up with two HIP
#                              runtimes, the second of which finds "No HIP GPUs" (seen when a test file that imports
#                              torch only inside a test ran after GPU tests had used the library)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def embedded():
    with open(os.path.join(GOLDEN, "morna_embedded_fixtures.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def embedded_mats():
    return dict(np.load(os.path.join(GOLDEN, "embedded_feature_matrices.npz")))


def angular64(X, i):
    """fp64 angular distance^2 (2 - 2cos) from row i to every row."""
    X = np.asarray(X, dtype=np.float64)
    n2 = (X * X).sum(1)
    pq = X @ X[i]
    ppqq = n2 * n2[i]
    with np.errstate(divide="ignore", invalid="ignore"):
        d = np.where(ppqq > 0, 2.0 - 2.0 * pq / np.sqrt(ppqq), 2.0)
    return d


def assert_tie_aware_order(got, expected, d, tol=1e-6):
    """got/expected: id sequences; d: exact distance of every id.  Equal up to
    permutations inside groups of tied distances."""
    got = list(got)
    expected = list(expected)
    assert sorted(got) == sorted(expected), (got, expected)
    dg = np.array([d[j] for j in got])
    de = np.array([d[j] for j in expected])
    assert np.all(np.diff(dg) >= -tol), (got, dg)
    assert np.allclose(dg, de, atol=tol, rtol=0), (got, expected, dg, de)
