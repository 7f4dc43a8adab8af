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


def pytest_generate_tests(metafunc):
    # traversal tests that do not name a kernel run three times: with the library's own choice ("auto":
    # small batches go to the one-per-wavefront kernel), with the four-per-wavefront kernel forced, and
    # with that kernel on the grouped visited table (the library computes a graph-locality layout for
    # whatever graph the test installed); kernels that have no grouped variant keep the hash table
    if "trav_mode" in metafunc.fixturenames:
        metafunc.parametrize("trav_mode", ["auto", "trav4", "trav4-grouped", "trav4-local"], indirect=True)


@pytest.fixture
def trav_mode(request, monkeypatch):
    if request.param in ("trav4", "trav4-grouped", "trav4-local"):
        monkeypatch.setenv("RADHIP_TRAV", "4")
    if request.param == "trav4-grouped":
        monkeypatch.setenv("RADHIP_TABLE", "group")
    elif request.param == "trav4-local":      # the bucket table hashed by the layout id (round 4)
        monkeypatch.setenv("RADHIP_TABLE", "local")
    else:
        monkeypatch.delenv("RADHIP_TABLE", raising=False)
    return request.param
