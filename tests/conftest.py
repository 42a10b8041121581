import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# the reference's multi-thread reductions are not reproducible run to run;
# the oracle mirrors its OpenMP structure, so pin it to one thread in tests
os.environ.setdefault("OMP_NUM_THREADS", "1")

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from oracle.pyoracle import Ref
    if not Ref.available():
        pytest.skip("oracle/_ref not built (reference tree absent)")
    return Ref()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
