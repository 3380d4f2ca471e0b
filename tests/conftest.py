import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib

    return oracle_lib.Oracle()


@pytest.fixture(scope="session")
def gpu_ctx_factory():
    """Returns a function making Contexts; fails (does not skip) if the HIP library is missing."""
    import c_lwe_snarks_amd as mf

    made = []

    def make(params=mf.DEBUG):
        c = mf.Context(params, 0)
        made.append(c)
        return c

    yield make
    for c in made:
        c.close()
