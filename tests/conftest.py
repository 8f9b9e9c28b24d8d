"""pytest configuration: registers the ``gpu`` marker and puts the repo root on sys.path."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _torch_thread_count_does_not_leak():
    """The fp32 oracle runs on torch's CPU kernels, whose summation order (and so the last bits of every expected value) depends
    on the intra-op thread count.  A test that changes it (tests/test_sf_agent_gpu.py pins 8 threads for a big oracle step) must
    not change what later tests compare against: the stated tolerances sit a few fp32 ulps above the oracle's own noise."""
    import torch
    n = torch.get_num_threads()
    yield
    if torch.get_num_threads() != n:
        torch.set_num_threads(n)
