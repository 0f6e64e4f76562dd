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


def pytest_terminal_summary(terminalreporter):
    """How many tests of this run were checked against the reference's own code (oracle/_ref) -- a number that silently
    drops to zero is what GSR_REQUIRE_REF=1 turns into failures."""
    try:
        import helpers
    except ImportError:
        return
    n = len(helpers.REF_BACKED)
    req = os.environ.get("GSR_REQUIRE_REF") == "1"
    line = f"reference-backed tests (oracle/_ref): {n} ran" + (" [GSR_REQUIRE_REF=1]" if req else "")
    # a GPU run in which tests that need the reference build were SKIPPED: say so where the result line is read
    skipped = [r for r in terminalreporter.stats.get("skipped", []) if "oracle/_ref" in str(getattr(r, "longrepr", ""))]
    if skipped:
        line += (f"; {len(skipped)} SKIPPED because oracle/_ref is missing -- the comparison against the reference's own kernels "
                 "did not run (GSR_REQUIRE_REF=1 turns that into failures)")
    terminalreporter.write_line(line)
