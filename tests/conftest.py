import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (ctypes over libmi355rt.so).  Built on demand; never falls back to the CPU."""
    p = graft.load_package()
    if not os.path.exists(p.LIB_PATH):
        p.build()
    return p


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle -- the checker (oracle/rt_oracle.h)."""
    return graft.load_oracle()


def scene_path(name):
    return os.path.join(ROOT, "scenes", name + ".yml")


def compare(got, want, rtol=1e-5, atol=1e-7):
    """Per-channel comparison used by every parity test: a channel passes if it is within `rtol` RELATIVE
    (BASELINE.json north_star: 1e-5) or within `atol` absolute (1e-7: far below one FP32 ulp of a colour in
    [0, 1]; only there so that channels that should be exactly 0 do not divide by zero)."""
    import numpy as np
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    diff = np.abs(got - want)
    rel = diff / np.maximum(np.maximum(np.abs(got), np.abs(want)), 1e-300)
    bad = (rel > rtol) & (diff > atol)
    return dict(n_bad=int(bad.sum()), n_bad_pixels=int(bad.any(axis=-1).sum()) if bad.ndim >= 1 else int(bad),
                max_rel=float(rel[diff > atol].max()) if (diff > atol).any() else 0.0, max_abs=float(diff.max()) if diff.size else 0.0,
                identical=bool(np.array_equal(got, want)))
