import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "cilrs-autonomous-driving-carla_amd")
for p in (PKG_DIR, os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _cpu_threads_fit_the_box():
    """The CPU oracle runs inside the GPU tests: size torch's thread pool to the cores this job
    may really use (cilrs_mi355/hostinfo.py) -- the affinity mask of a shared GPU box is the whole
    machine and oversubscribing the job's CPU share makes every oracle step crawl."""
    try:
        import torch
        from cilrs_mi355.hostinfo import usable_cores
        torch.set_num_threads(usable_cores())
    except Exception:
        pass
    yield
