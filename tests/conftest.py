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


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure). Built on demand with gcc."""
    from oracle import oracle as O

    O.build()
    O.lib()
    return O


@pytest.fixture(scope="session")
def hvd():
    import hvd_amd

    return hvd_amd


@pytest.fixture(scope="session")
def gpu(hvd):
    """Initialised library on HIP device 0 -- fails (not skips) if the HIP extension is unusable."""
    from hvd_amd import _lib

    _lib.init(0)
    return _lib


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def pairs_as_tuples(recs):
    return [(int(r["i"]), int(r["j"]), int(r["dist"])) for r in recs]
