"""The overlapped data-parallel schedule (SH exchange + Adam on a side stream under the next step's geometry stages)
gives bit-identical parameters to the plain synchronous schedule.  Two ranks share the one GPU of the test box (gloo between
them: RCCL needs one device per rank), so this checks the schedule's correctness, not its speed."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_overlapped_schedule_is_bit_identical():
    port = 29600 + (os.getpid() % 300)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dp_overlap_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "DP_OVERLAP_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
