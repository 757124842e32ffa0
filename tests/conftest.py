import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """(re)compile the extension whenever its sources changed (hipcc cross-compiles without a GPU; build() compares
    a source hash stored next to the library and returns at once when nothing changed).  Building is not a
    fallback — without hipcc the engine tests fail loudly."""
    try:
        from lp_mp_amd import build as B
        B.build()
    except Exception as e:                                   # reported by the tests that need the library
        print(f"conftest: could not build the HIP extension: {e}", file=sys.stderr)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
