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
    # The CPU oracles are torch code: let torch use the CPUs this process really has (a GPU box gives a 16-CPU share of a much
    # bigger host; `os.cpu_count()` threads on that share spin against each other - an oracle-heavy test then takes minutes
    # instead of seconds, which is what a test that "hung" once in round 3 most likely was)
    import torch
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:       # noqa: BLE001
        n = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(n, int(os.environ.get("GSR_TEST_THREADS", "16")))))


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
