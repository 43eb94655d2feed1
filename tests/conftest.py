import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(params=[0, 1, 2, 3, 6], ids=lambda v: "waves%d" % v)
def waves_per_board(request):
    """Runs a GPU test once per launch shape of the kernels (wavefronts that share one board;
    0 = chosen per launch from the batch size).  Results must not depend on it."""
    if request.param:
        os.environ["GOENV_WAVES_PER_BOARD"] = str(request.param)
    else:
        os.environ.pop("GOENV_WAVES_PER_BOARD", None)
    yield request.param
    os.environ.pop("GOENV_WAVES_PER_BOARD", None)
