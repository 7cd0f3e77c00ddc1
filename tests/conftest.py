import os
import sys

import pytest

# Small CPU-only containers (the 8-core build box): eight spinning OpenMP workers per parallel region turn the host-logic
# tests -- thousands of tiny torch ops each -- into a crawl once the process has been alive for a minute (measured here:
# the 60 CPU tests in 40+ minutes with the defaults, 9 with passive waiting, 5 with four passive threads; the same
# tests alone take 30 s).  Set before torch is imported; the GPU box (128 cores, minutes of fp32 oracle) keeps the defaults.
if (os.cpu_count() or 1) <= 16:
    os.environ.setdefault("OMP_NUM_THREADS", "4")
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: minutes of CPU oracle (still part of -m gpu)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
