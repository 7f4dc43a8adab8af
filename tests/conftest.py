import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure only)."""
    from oracle import rad_oracle
    rad_oracle.build()
    return rad_oracle


@pytest.fixture(scope="session")
def gpu():
    """Fails loudly when the HIP library or the GPU is missing: no fallback."""
    from rad_amd import _lib
    _lib.lib()
    n = _lib.device_count()
    assert n > 0, "no HIP device visible — gpu-marked tests must run on an MI355X"
    return n
