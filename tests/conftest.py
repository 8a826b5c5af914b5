import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import pkgload
    p = pkgload.load_package()
    if not os.path.exists(p.lib_path()):
        p.build_all()
    return p


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    from oracle.oracle import Reference, have_reference
    if not have_reference():
        pytest.skip("oracle/_ref/native.so not built (needs /root/reference)")
    return Reference()


@pytest.fixture(scope="session")
def backend(pkg):
    """The HIP backend.  GPU tests must run on the real library: no skip, no fallback."""
    be = pkg.Backend()
    if be.device_count() < 1:
        pytest.fail("no HIP device visible: GPU tests need a GPU (the backend has no CPU fallback)")
    return be
