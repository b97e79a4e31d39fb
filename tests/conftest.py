import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def _have_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

# The GPU tests call `engine.capture()` + `engine.replay(n)` to hold hipGraph replays of the decode step to the same
# tokens as eager steps (tag counters, arrival counters and device-side positions must survive a replayed graph). The
# product default is the graph too (runtime/engine.py LAUNCH); pinned here so that a WOQ_ENGINE_LAUNCH=eager left in the
# environment cannot take the graph path out of the suite. tests/test_gpu_engine.py::test_eager_bursts_equal_graph_replays
# covers the eager burst against it.
import os  # noqa: E402

os.environ.setdefault("WOQ_ENGINE_LAUNCH", "graph")
