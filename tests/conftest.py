import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (oracle/vfsms_oracle.c) -- the checker, never the thing under test."""
    from oracle import oracle as O
    O.build()
    O.lib()
    return O


@pytest.fixture(scope="session")
def engine():
    """The product: libvfsms.so on cuda:0.  Fails loudly when the extension or the GPU is missing."""
    import imagestitch_amd as isa
    lib = isa.load_library()
    assert lib.vfsms_device_count() > 0, "gpu tests need a visible MI355X"
    eng = isa.Engine(0)
    yield eng
    eng.close()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
