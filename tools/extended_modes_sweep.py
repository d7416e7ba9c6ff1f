#!/usr/bin/env python3
"""Extended bit-equality sweep over the forward modes and binning forms (GPU box, repo root):
    python tools/extended_modes_sweep.py [first] [count]
Per seed a scene of tests/test_sweep_gpu.py's generator, rendered forward + backward through (a) the blocking forward with the
global depth sort ("sync" + global binning: the published structure), (b) the default - speculative, verified, tile-local binning
- (c) the unverified mode, each with a workspace pool that has never seen the shape (so the capacity guess, the verify and, when
the guess was too small, the re-render all take part).  Images, radii and every gradient must be equal bit for bit."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-slam_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import diff_gaussian_rasterization as dgr  # noqa: E402
from diff_gaussian_rasterization import _workspace as ws  # noqa: E402
from helpers import run_hip, upstream_grads  # noqa: E402
from test_sweep_gpu import _case  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad, rerendered, t0 = [], 0, time.time()
old_mode, old_bin, old_min = ws.forward_mode(), ws._BINNING, ws.MIN_CAPACITY
ws.MIN_CAPACITY = 256                       # (small scenes: let the first-frame guess be wrong sometimes)
try:
    for seed in range(first, first + count):
        raw, cam, deg, mode, aa, bg, sm = _case(seed)
        gc, gd = upstream_grads(cam.image_height, cam.image_width, seed=seed)
        outs = []
        try:
            for fwd, binning in (("sync", "global"), ("exact", "tile"), ("async", "tile"), ("exact", "global")):
                dgr.set_forward_mode(fwd)
                ws._BINNING = binning
                pool = ws.pool(torch.device("cuda", 0))
                pool.forget_estimates()
                before = pool.stats["rerendered_frames"]
                outs.append(run_hip(raw, cam, deg, bg, mode=mode, antialiasing=aa, scale_modifier=sm, gc=gc, gd=gd))
                rerendered += pool.stats["rerendered_frames"] - before
            a = outs[0]
            for b in outs[1:]:
                for k in ("color", "radii", "invdepth"):
                    assert torch.equal(a[k], b[k]), k
                for k in a["grads"]:
                    assert torch.equal(a["grads"][k], b["grads"][k]), ("grad", k)
        except Exception as e:      # noqa: BLE001
            bad.append(seed)
            print(f"seed {seed}: {type(e).__name__}: {str(e)[:300]}", flush=True)
        if (seed - first) % 50 == 49:
            print(f"... {seed - first + 1} cases, {len(bad)} failures, {rerendered} re-rendered frames, {time.time() - t0:.0f} s", flush=True)
finally:
    dgr.set_forward_mode(old_mode)
    ws._BINNING, ws.MIN_CAPACITY = old_bin, old_min
print(f"forward-mode sweep: seeds {first}..{first + count - 1}: {count - len(bad)} bit-identical in all four combinations "
      f"({rerendered} frames re-rendered on the way), {len(bad)} failed {bad}")
sys.exit(1 if bad else 0)
