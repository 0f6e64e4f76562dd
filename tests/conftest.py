import os
import sys

import pytest

# Several tests import the reference's own Python from /root/reference (read-only input of this work): never let the
# interpreter -- this process or a subprocess a test starts -- drop __pycache__/*.pyc files into that tree.
sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import cpu

    cpu.build()
    return cpu
