import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def mi_ctx():
    """One libmi355interp context on cuda:0 for the whole GPU session (fails loudly without a GPU)."""
    import armadillocudalinearinterpolation_amd as mi
    ctx = mi.Context(0)
    yield ctx
    ctx.close()
