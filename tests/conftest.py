import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-segmentation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "multigpu: needs >= 2 MI355X in one node; DESELECTED (not skipped) elsewhere")


def pytest_collection_modifyitems(config, items):
    """Tests that need two GPUs of one node are deselected on smaller boxes: a one-GPU run reports no skips."""
    multi = [it for it in items if it.get_closest_marker("multigpu")]
    if not multi:
        return
    try:
        import torch
        n = torch.cuda.device_count()
    except Exception:       # noqa: BLE001
        n = 0
    if n < 2:
        config.hook.pytest_deselected(items=multi)
        items[:] = [it for it in items if not it.get_closest_marker("multigpu")]


@pytest.fixture(scope="session")
def out_dir():
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    return d
