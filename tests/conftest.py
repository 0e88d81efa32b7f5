import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "experimental: parity cases of the kernel forms of `make EXPERIMENTAL=1` (bit-exact, measured slower than "
                            "the defaults); run them with DRPRG_HIP_LIB=build/exp/libdrprg_hip.so pytest -m 'gpu and experimental'")
    from util import ensure_built
    ensure_built()


@pytest.fixture(scope="session")
def oracle():
    from util import Oracle
    return Oracle()


def pytest_collection_modifyitems(config, items):
    """the experimental cases need the library of `make EXPERIMENTAL=1` (DRPRG_HIP_LIB=build/exp/libdrprg_hip.so): skipped on the default one"""
    from drprg_amd import _lib
    if _lib.lib.drprg_hip_experimental():
        return
    skip = pytest.mark.skip(reason="library built without EXPERIMENTAL=1")
    for item in items:
        if "experimental" in item.keywords:
            item.add_marker(skip)
