import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the native artefacts exist (no-op when already built)."""
    import __graft_entry__ as g
    if not (os.path.exists(os.path.join(ROOT, "khronos_amd", "lib", "libkhronos_amd.so"))
            and os.path.exists(os.path.join(ROOT, "khronos_amd", "lib", "libkhr_synth.so"))
            and os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so"))):
        g.build()
