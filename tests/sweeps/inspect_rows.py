#!/usr/bin/env python3
"""Which Gaussians carry a seed's gradient error?  python tests/sweeps/inspect_rows.py <seed> [tensor]   (GPU box, repo root)
Prints, for the rows of `tensor` (default means3D) with the largest |hip - float64 oracle|, the row's gradients in the HIP path, the
fp32 oracle and the float64 oracle, and what the projection did with the Gaussian (radius, depth, position relative to the frustum
clamp, determinant of the dilated 2-D covariance)."""
import os, sys, math
ROOT = os.getcwd()
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-slam_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from helpers import run_hip, run_oracle, upstream_grads, leaf_inputs
from test_sweep_gpu import _case
torch.set_num_threads(16)
seed = int(sys.argv[1]); name = sys.argv[2] if len(sys.argv) > 2 else "means3D"
raw, cam, deg, mode, aa, bg, sm = _case(seed)
gc, gd = upstream_grads(cam.image_height, cam.image_width, seed=seed)
kw = dict(mode=mode, antialiasing=aa, scale_modifier=sm, gc=gc, gd=gd)
ref = run_oracle(raw, cam, deg, bg, torch.float64, **kw)
r32 = run_oracle(raw, cam, deg, bg, torch.float32, **kw)
out = run_hip(raw, cam, deg, bg, **kw)
g, g32, g64 = out["grads"][name].double().cpu(), r32["grads"][name].double(), ref["grads"][name].double()
err = (g - g64).abs().flatten(1).max(dim=1).values
print(f"seed {seed} {name}: max|g64| {float(g64.abs().max()):.3e}  total rel-L2 hip {float((g-g64).norm()/g64.norm()):.2e} oracle32 {float((g32-g64).norm()/g64.norm()):.2e}")
inp = leaf_inputs(raw, torch.float64, "cpu", mode)
xyz = inp["means3D"].detach()
V = cam.world_view_transform.double()
t = torch.cat([xyz, torch.ones(len(xyz), 1, dtype=torch.float64)], 1) @ V
tanx, tany = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
for i in err.argsort(descending=True)[:6].tolist():
    tx, ty, tz = (float(v) for v in t[i, :3])
    print(f" row {i}: err {float(err[i]):.3e}  radius {int(out['radii'][i])}  t=({tx:.4f},{ty:.4f},{tz:.4f})  x/z over 1.3tan: {abs(tx/tz)/(1.3*tanx):.6f}  y/z: {abs(ty/tz)/(1.3*tany):.6f}")
    print("    hip     ", [f"{float(v):+.5e}" for v in g[i].flatten()[:4]])
    print("    oracle32", [f"{float(v):+.5e}" for v in g32[i].flatten()[:4]])
    print("    oracle64", [f"{float(v):+.5e}" for v in g64[i].flatten()[:4]])
    for other in ("means2D", "opacities", "scales"):
        a, b = out["grads"][other].double().cpu()[i].flatten()[:3], ref["grads"][other].double()[i].flatten()[:3]
        print(f"    {other:9s} hip {[f'{float(v):+.4e}' for v in a]}  oracle64 {[f'{float(v):+.4e}' for v in b]}")
    print("    scale (activated)", [f"{float(v):.3e}" for v in inp["scales"][i].detach()], "opacity", f"{float(inp['opacities'][i].detach()):.4f}")
