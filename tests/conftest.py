import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# "1e-5 relative fp32" of BASELINE.json north_star, defined per tensor as in
# max-abs error relative to max-abs reference AND relative L2.
REL_TOL = 1e-5


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


class Fixture:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.meta = json.loads(str(z["meta"]))
        self.arrays = {k: z[k] for k in z.files if k != "meta"}

    def __getitem__(self, k):
        return self.arrays[k]

    def __contains__(self, k):
        return k in self.arrays


def load_golden(name):
    return Fixture(name)


def rel_err(a, b):
    """(max-abs error / max-abs ref, L2 error / L2 ref) with b the reference."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    den_max = max(np.abs(b).max(), 1e-30) if b.size else 1.0
    den_l2 = max(np.linalg.norm(b), 1e-30) if b.size else 1.0
    if not a.size:
        return 0.0, 0.0
    return np.abs(a - b).max() / den_max, np.linalg.norm(a - b) / den_l2


def assert_close(a, b, tol=REL_TOL, what=""):
    e_max, e_l2 = rel_err(a, b)
    assert e_max <= tol and e_l2 <= tol, f"{what}: rel-to-max {e_max:.3e}, rel-L2 {e_l2:.3e} > {tol}"


@pytest.fixture(scope="session")
def golden():
    return load_golden
