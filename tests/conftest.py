"""pytest configuration: the `gpu` marker and shared fixtures.

`-m "not gpu"` runs on any CPU box (oracle vs goldens, host logic, ABI export
checks, gloo world_size-2 sharding).  `-m gpu` needs a real MI355X and drives
the HIP engine through its C-ABI.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (HIP engine through the C-ABI)")


@pytest.fixture(scope="session")
def orc():
    from oracle.pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _gpu_tests_run_with_the_host_path_off(request):
    """The engine's own host data path (csrc/uaes_host.c) is opt-in.  Every `-m gpu` test must find the default policy
    (0, 0, 0) -- nothing ever runs on the host, a missing GPU fails loudly -- so that no GPU parity test can be passed
    by host code.  (tests/test_host_path.py forces the host path in CPU tests of its own.)"""
    if request.node.get_closest_marker("gpu") is not None:
        import micro_aes_amd as uaes
        assert uaes.host_policy() == (0, 0, 0), "a GPU parity test must not run with the host data path switched on"
        for name in ("UAES_HOST_MAX", "UAES_HOST_CHAINS", "UAES_HOST_FALLBACK", "UAES_HOST_POLICY"):
            assert not os.environ.get(name), "%s is set: GPU parity tests run with the host path off" % name
    yield
