import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests fail loudly (not skip) when selected on a box without a device; they are simply
    deselected by `-m "not gpu"` on CPU."""
    return


@pytest.fixture(scope="session")
def golden():
    from tests.util import Golden
    return Golden()
