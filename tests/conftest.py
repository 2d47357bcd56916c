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
