import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
CASES = os.path.join(ROOT, "data", "BenchmarkCases")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the native pieces exist (product .so, host maths .so, oracle .so)."""
    import __graft_entry__ as g
    from automatedvaletparking_amd import _native
    if not (os.path.exists(_native.LIB_PATH) and os.path.exists(_native.HOSTMATH_PATH)
            and os.path.exists(os.path.join(ROOT, "oracle", "libavp_oracle.so"))):
        g.build()
    yield


def gold(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def case_map_from_gold(k, g1=None):
    """Map of BenchmarkCase k rebuilt from the stored golden occupancy cells."""
    from automatedvaletparking_amd import costmap
    g1 = g1 if g1 is not None else gold("g1_costmaps.npz")
    case = costmap.Case.read(os.path.join(CASES, f"Case{k}.csv"))
    return costmap.Map.from_cells(case, g1[f"c{k}_boundary"], int(g1[f"c{k}_nx"]), int(g1[f"c{k}_ny"]), g1[f"c{k}_cells"])


@pytest.fixture(scope="session")
def cfg():
    from automatedvaletparking_amd import config
    return config.default_config()


@pytest.fixture(scope="session")
def vehicle():
    from automatedvaletparking_amd import costmap
    return costmap.Vehicle()
