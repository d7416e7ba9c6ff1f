import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gaussian-splatting-slam_amd")
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _fresh_capacity_estimates(request):
    """The non-blocking forward sizes its binning state from earlier frames of the same (P, W, H): two tests that share a
    shape but not a scene would hand each other a wrong capacity (a frame beyond it drops instances).  Every GPU test starts
    with no estimates, as a fresh process would."""
    yield
    if "gpu" in request.keywords:
        import torch
        if torch.cuda.is_available():
            from diff_gaussian_rasterization import _workspace as ws
            for p in ws._pools.values():
                p.forget_estimates()
