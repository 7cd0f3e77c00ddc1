import os
import sys

import pytest

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


def pytest_collection_finish(session):
    """A `-m gpu` session on a GPU box: start the minutes-long CPU oracle runs of the full-size parity tests NOW, in child
    processes, so they run under the GPU tests that come first (tests/oracle_ahead.py)."""
    import torch

    if not torch.cuda.is_available():
        return
    import oracle_ahead

    wanted = {"test_baseline_config2_512_four_step_matches_oracle": "sd15_config2",
              "test_baseline_config5_768_eight_step_scale2_matches_oracle": "sd15_config5",
              "test_reference_only_mode_512_four_step_matches_oracle": "sd15_ref512",
              "test_sdxl_1024_four_step_matches_oracle": "sdxl_1024"}
    names = [wanted[it.name] for it in session.items if it.name in wanted and "skip" not in it.keywords]
    if len(session.items) > 40 and names:  # (a whole-suite run: a test picked by hand computes its oracle inline)
        oracle_ahead.start_all(sorted(set(names), key=["sdxl_1024", "sd15_config5", "sd15_config2", "sd15_ref512"].index))


def pytest_sessionfinish(session, exitstatus):
    try:
        import oracle_ahead

        oracle_ahead.stop_all()
    except Exception:
        pass
