import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# MIOpen keeps the result of its solver search ("find") for every convolution shape in a user database ON DISK, and the training tests
# run config #5's shapes under torch's deterministic mode, where the search is restricted to deterministic solvers (for some weight
# gradients: the naive reference kernel).  Left in the default location those entries were picked up by the NEXT process on the box -
# round 5 measured bench.py's train_full child at 634 ms per step instead of 128 after a full test session (7 launches of
# naive_conv_..._wrw of 68 ms each per step).  The test session therefore keeps its MIOpen databases in a directory of its own.
# One stable directory per purpose (a fresh mkdtemp per session piled up under /tmp and redid every search).
if "MIOPEN_USER_DB_PATH" not in os.environ:
    _miopen_dir = os.path.join(os.path.expanduser("~"), ".cache", "lav_amd", "miopen_tests")
    try:
        os.makedirs(_miopen_dir, exist_ok=True)
    except OSError:
        import tempfile
        _miopen_dir = os.path.join(tempfile.gettempdir(), "lav_amd_miopen_tests")
        os.makedirs(_miopen_dir, exist_ok=True)
    os.environ["MIOPEN_USER_DB_PATH"] = _miopen_dir
    os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", _miopen_dir)


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
