import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _gpu_available():
    try:
        from genometools_amd import _lib
        return _lib.load().gtamd_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests must fail loudly, not skip, when the HIP library is missing on
    a box that was asked to run them."""
    from genometools_amd import _lib
    lib = _lib.load()
    assert lib.gtamd_device_count() > 0, "no HIP device visible"
    return lib
