#!/usr/bin/env python3
"""Extended run of tests/test_sweep_gpu.py's seeded sweep (GPU box, repo root):  python tests/sweeps/extended_sweep.py [first] [count]
Every case forward + backward against the float64 oracle with the tolerances of tests/test_parity_gpu.py; failures are listed, not
raised, so one run shows all of them.  (The test suite keeps seeds 0..23; this is for spare GPU minutes.)

On images of a few thousand pixels the suite's "99.99 % of the pixels within 2e-5" leaves room for no pixel at all, so a case whose
ONLY deviation is at most 3 pixels off by less than one minimal contribution (alpha = 1/255: 4e-3 of the value range; a decision
taken differently at a discontinuity - alpha >= 1/255, T < 1e-4, power > 0 - or fp32 rounding of a large inverse depth) is counted
as a "flip", not as a failure, provided its gradients pass."""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-slam_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from helpers import run_hip, run_oracle, upstream_grads  # noqa: E402
from test_parity_gpu import check_forward, check_grads  # noqa: E402
from test_sweep_gpu import _case  # noqa: E402

COV = "--cov" in sys.argv
_a = [a for a in sys.argv[1:] if a != "--cov"]
first = int(_a[0]) if len(_a) > 0 else 24
count = int(_a[1]) if len(_a) > 1 else 200
torch.set_num_threads(16)
bad, flips, t0 = [], [], time.time()
for seed in range(first, first + count):
    try:
        raw, cam, deg, mode, aa, bg, sm = _case(seed)
        gc, gd = upstream_grads(cam.image_height, cam.image_width, seed=seed)
        cov = COV and seed % 2 == 0            # --cov: every other case hands in a precomputed 3-D covariance instead of scale / rotation
        ref = run_oracle(raw, cam, deg, bg, torch.float64, mode=mode, antialiasing=aa, scale_modifier=sm, gc=gc, gd=gd,
                         cov_precomp=cov)
        out = run_hip(raw, cam, deg, bg, mode=mode, antialiasing=aa, scale_modifier=sm, gc=gc, gd=gd, cov_precomp=cov)
        try:
            check_forward(out, ref)
        except AssertionError as e:
            n_bad, worst = 0, 0.0
            for k in ("color", "invdepth"):
                d = (ref[k].double() - out[k].double()).abs() / max(1.0, float(ref[k].abs().max()))
                px = (d > 2e-5).reshape(-1, d.shape[-2], d.shape[-1]).any(0)
                n_bad, worst = max(n_bad, int(px.sum())), max(worst, float(d.max()))
            if not (torch.equal(ref["radii"].long(), out["radii"].long()) and n_bad <= 3 and worst <= 4e-3):
                raise
            flips.append(seed)
        check_grads(out, ref)
    except Exception as e:      # noqa: BLE001
        bad.append(seed)
        print(f"seed {seed}: {type(e).__name__}: {str(e)[:300]}", flush=True)
        if os.environ.get("SWEEP_TRACE"):
            traceback.print_exc()
    if (seed - first) % 25 == 24:
        print(f"... {seed - first + 1} cases, {len(bad)} failures, {time.time() - t0:.0f} s", flush=True)
print(f"extended sweep: seeds {first}..{first + count - 1}: {count - len(bad)} passed ({len(flips)} of them with <= 3 flipped pixels: "
      f"{flips}), {len(bad)} failed {bad}")
sys.exit(1 if bad else 0)
