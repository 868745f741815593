import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def _built(path):
    return os.path.exists(os.path.join(ROOT, path))


@pytest.fixture(scope="session", autouse=True)
def _ensure_built():
    """CPU-side libraries are built on demand (seconds); the HIP library is built by
    __graft_entry__.build() / make and travels to the GPU box with the snapshot."""
    if not (_built("luisarender_amd/lib/liblrhost.so") and _built("oracle/liboracle.so")):
        import subprocess
        subprocess.check_call(["make", "-C", ROOT, "host", "oracle"], stdout=subprocess.DEVNULL)
    yield
