import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle is plain C and builds in a second; the HIP library is built by
    # __graft_entry__.build() and travels to the GPU box as an in-tree .so
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


@pytest.fixture(scope="session")
def oracle():
    from oracle_binding import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def pqv():
    """The product package; GPU tests fail loudly if the HIP library or a device is missing."""
    import pq_vector_amd
    from pq_vector_amd import _ffi
    _ffi.lib()
    if pq_vector_amd.device_count() < 1:
        pytest.fail("no HIP device: pq_vector_amd has no CPU fallback")
    return pq_vector_amd
