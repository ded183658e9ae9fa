import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure the in-tree native library exists (cross-compiles here without a GPU)."""
    import __graft_entry__ as g

    g.build()
    return True


@pytest.fixture(scope="session")
def gpu_ctx(built):
    from qml_cutensornet_amd import engine

    ctx = engine.Context(0)
    yield ctx
    ctx.close()
