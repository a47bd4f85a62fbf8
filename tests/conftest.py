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
    """Build the native artefacts; build() is mtime-gated, so this is a no-op when they are newer than their sources
    (lib/ is git-ignored: running against stale binaries after a source edit must not be possible)."""
    import __graft_entry__ as g
    g.build()


@pytest.fixture(params=["fast", "exact"])
def arith(request):
    """Runs a parity test under both arithmetic modes of the voxel update (khr_config.exact_arithmetic): `fast` = the
    relaxed option (decisions exact, values within TOL), `exact` = the product default: values bit-identical to the oracle as well."""
    import common
    prev = common.EXACT
    common.EXACT = 1 if request.param == "exact" else 0
    yield request.param
    common.EXACT = prev
