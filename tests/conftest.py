import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU check (opt-in)")


@pytest.fixture(params=["parallel fixed point", "sequential"])
def resolver(request):
    """The whole-loop projection searches under both resolvers (k_resolve_par / k_resolve): same results required."""
    from orb_slam2_e_amd._lib import lib
    L = lib()
    prev = L.orbm_debug_force_sequential_resolver(1 if request.param == "sequential" else 0)
    yield request.param
    L.orbm_debug_force_sequential_resolver(prev)
