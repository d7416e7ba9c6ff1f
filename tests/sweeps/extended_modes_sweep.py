#!/usr/bin/env python3
"""Extended bit-equality sweep over the forward modes and binning forms (GPU box, repo root):
    python tests/sweeps/extended_modes_sweep.py [first] [count]
Per seed a scene of tests/test_sweep_gpu.py's generator, rendered forward + backward through (a) the blocking forward with the
global depth sort ("sync" + global binning: the published structure), (b) the default - speculative, verified, tile-local binning
- (c) the unverified mode, each with a workspace pool that has never seen the shape (so the capacity guess, the verify and, when
the guess was too small, the re-render all take part).  Images, radii and every gradient must be equal bit for bit."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-slam_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

torch.set_num_threads(16)      # (the box's CPU share; torch's default there is 128 threads on a 16-CPU quota)

import diff_gaussian_rasterization as dgr  # noqa: E402
from diff_gaussian_rasterization import _workspace as ws  # noqa: E402
from helpers import run_hip, upstream_grads  # noqa: E402
from test_sweep_gpu import _case  # noqa: E402

LONG = "--long" in sys.argv          # many Gaussians on a small image: tile lists of 1 000 - 9 000 entries (the per-tile ordering's
args = [a for a in sys.argv[1:] if a != "--long"]   # second launch, <= 4096, and its in-memory path beyond the LDS capacity)
first = int(args[0]) if len(args) > 0 else 0
count = int(args[1]) if len(args) > 1 else 200


def _long_case(seed):
    import math
    from scene_utils import make_gaussians, look_at_camera
    g = torch.Generator().manual_seed(8000 + seed)

    def u(a, b):
        return a + (b - a) * float(torch.rand((), generator=g))
    P, W, H, deg = int(u(1500, 9000)), int(u(17, 70)), int(u(17, 60)), seed % 4
    raw = make_gaussians(P, deg, seed=8100 + seed, scale_factor=u(0.5, 3.0))
    raw.opacity += u(-4.0, 0.0)               # mostly faint: the lists stay long instead of saturating early
    th, ph = u(0, 2 * math.pi), u(-1.0, 1.0)
    eye = (4.0 * math.cos(th) * math.cos(ph), 4.0 * math.sin(th) * math.cos(ph), 4.0 * math.sin(ph))
    cam = look_at_camera(eye, (0.0, 0.0, 0.0), (0, 0, 1), u(0.5, 0.9), W, H)
    return raw, cam, deg, ("sh", "dc", "colors")[(seed // 4) % 3], bool(seed % 2), torch.rand(3, generator=g), 1.0

bad, rerendered, longest, t0 = [], 0, 0, time.time()
old_mode, old_bin, old_min = ws.forward_mode(), ws._BINNING, ws.MIN_CAPACITY
ws.MIN_CAPACITY = 256                       # (small scenes: let the first-frame guess be wrong sometimes)
try:
    for seed in range(first, first + count):
        raw, cam, deg, mode, aa, bg, sm = _long_case(seed) if LONG else _case(seed)
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
                pool.poll(wait=True)
                longest = max([longest] + list(pool.longest_list.values()))
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
            print(f"... {seed - first + 1} cases, {len(bad)} failures, {rerendered} re-rendered frames, longest tile list so far "
                  f"{longest}, {time.time() - t0:.0f} s", flush=True)
finally:
    dgr.set_forward_mode(old_mode)
    ws._BINNING, ws.MIN_CAPACITY = old_bin, old_min
print(f"forward-mode sweep: seeds {first}..{first + count - 1}: {count - len(bad)} bit-identical in all four combinations "
      f"({rerendered} frames re-rendered on the way), {len(bad)} failed {bad}")
sys.exit(1 if bad else 0)
