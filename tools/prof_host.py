#!/usr/bin/env python3
"""Host-side profile of the training loop where the host is what bounds it (BASELINE configs[0]: 10 k Gaussians, 256x256):
    python tools/prof_host.py [bench.py arguments]        (GPU box, repo root)
cProfile of bench.main(), 35 entries by internal time.  Round 3: no single hot spot - Trainer.step costs ~0.35 ms of Python,
ctypes and autograd bookkeeping per step spread over the forward, the loss, the backward and the optimizer hand-off."""
import cProfile
import io
import os
import pstats
import sys

sys.argv = ["bench.py"] + (sys.argv[1:] or ["--config", "1", "--gpus", "1", "--steps", "300", "--warmup", "20", "--no-cpu-baseline",
                                            "--no-kernel-profile"])
sys.path.insert(0, os.getcwd())
import bench  # noqa: E402

pr = cProfile.Profile()
pr.enable()
try:
    bench.main()
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
print(s.getvalue()[:8000])
