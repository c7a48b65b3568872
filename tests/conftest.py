import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


SHIPPED = {}      # state of the in-tree libmijpeg.so as it ARRIVED, before any fixture could rebuild it


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    try:
        from nvjpeg_imagecompressor_amd import build as B
        SHIPPED.update(present=os.path.exists(B.LIB), stale=B.needs_build(), tree_hash=B.source_hash())
    except Exception as e:       # noqa: BLE001 -- reported by the test that reads this
        SHIPPED.update(error=repr(e))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def mij():
    """The product package with its HIP library built (build works without a GPU)."""
    from nvjpeg_imagecompressor_amd import build as B
    B.build()
    import nvjpeg_imagecompressor_amd as pkg
    return pkg
