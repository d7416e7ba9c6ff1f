#!/usr/bin/env python3
"""Why did a seed of tests/sweeps/extended_sweep.py miss the tolerance?  python tests/sweeps/sweep_inspect.py <seed> [<seed> ...]   (GPU box)
Per seed: forward deviations of the HIP path and of the fp32 oracle against the float64 oracle, and per gradient tensor the
relative L2 error of (a) the HIP path, (b) the fp32 oracle, (c) the float64 oracle itself after moving every opacity logit by
+-1e-6.  (a) == (b): fp32 rounding; (c) of the same size as (a): the scene sits on a discontinuity (alpha >= 1/255, T < 1e-4,
power > 0, colour clamp) and an infinitesimal change flips a contribution - in float64 just as well."""
import os, sys, copy
ROOT = os.getcwd()
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-slam_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from helpers import run_hip, run_oracle, upstream_grads
from test_sweep_gpu import _case
from test_parity_gpu import rel_l2
torch.set_num_threads(16)
for seed in [int(a) for a in sys.argv[1:]] or (914, 1056, 1596, 1997, 2580):
    raw, cam, deg, mode, aa, bg, sm = _case(seed)
    gc, gd = upstream_grads(cam.image_height, cam.image_width, seed=seed)
    kw = dict(mode=mode, antialiasing=aa, scale_modifier=sm, gc=gc, gd=gd)
    ref = run_oracle(raw, cam, deg, bg, torch.float64, **kw)
    ref32 = run_oracle(raw, cam, deg, bg, torch.float32, **kw)
    out = run_hip(raw, cam, deg, bg, **kw)
    H, W = cam.image_height, cam.image_width
    print(f"seed {seed}: P {raw.xyz.shape[0]} {W}x{H} deg {deg} mode {mode} aa {aa} sm {sm:.2f}")
    for k in ("color", "invdepth"):
        d = (ref[k].double() - out[k].double()).abs(); d32 = (ref[k].double() - ref32[k].double()).abs()
        print(f"  {k}: hip n>2e-5 {int((d > 2e-5).sum())} max {float(d.max()):.2e} | oracle32 n> {int((d32 > 2e-5).sum())} max {float(d32.max()):.2e}")
    nc_h, nc_r = out.get("n_contrib"), ref.get("n_contrib")
    if nc_h is not None and nc_r is not None:
        print("  n_contrib differs at", int((nc_h.long() != nc_r.long()).sum()), "pixels; oracle32 vs 64:", int((ref32["n_contrib"].long() != nc_r.long()).sum()))
    # sensitivity of the float64 oracle itself: the same scene with every opacity logit moved by +-1e-6
    pert = []
    for eps in (1e-6, -1e-6):
        r2 = copy.deepcopy(raw)
        r2.opacity = r2.opacity + eps
        pert.append(run_oracle(r2, cam, deg, bg, torch.float64, **kw))
    for k, g_ref in ref["grads"].items():
        if g_ref.numel() == 0 or float(g_ref.abs().max()) == 0:
            continue
        e_h = rel_l2(out["grads"][k], g_ref); e_32 = rel_l2(ref32["grads"][k], g_ref)
        e_p = max(rel_l2(p["grads"][k], g_ref) for p in pert)
        flag = " <--" if e_h > 1e-4 else ""
        print(f"  grad {k:10s} hip {e_h:.2e}  oracle32 {e_32:.2e}  oracle64(opacity +-1e-6) {e_p:.2e}{flag}")
