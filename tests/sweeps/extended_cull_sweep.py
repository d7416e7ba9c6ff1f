#!/usr/bin/env python3
"""Extended sweep of the tile lists truncated by depth (GPU box, repo root):
    python tests/sweeps/extended_cull_sweep.py [first] [count]
Per seed a random small training run (both folded optimizers, SH 0..3, with / without densification, splat sizes from sparse to
heavily saturating, random margins down to ones that are far too tight) in forward mode "async" WITHOUT truncation against the same
run WITH it (Trainer.enable_tile_cull): 20 steps over 3 views, so every view's cut-offs are applied ~6 times; frames whose truncation
was too tight must flag themselves, be no-ops, and be run again.  The end state - parameters, both Adam moments, densification
statistics, iteration count - must be equal bit for bit."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-slam_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

torch.set_num_threads(16)

import diff_gaussian_rasterization as dgr  # noqa: E402
from diff_gaussian_rasterization import _workspace as ws  # noqa: E402
from gaussian_renderer import render, PipelineParams  # noqa: E402
from scene_utils import make_gaussians, fibonacci_cameras, GaussianModel, Trainer  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad, reruns, culled, flagged, t0 = [], 0, 0, 0, time.time()
old_mode = ws.forward_mode()
old_margin = os.environ.get("GSR_CULL_MARGIN")
dgr.set_forward_mode("async")
try:
    for seed in range(first, first + count):
        g = torch.Generator().manual_seed(7000 + seed)

        def u(a, b):
            return a + (b - a) * float(torch.rand((), generator=g))
        P, W, H, deg = int(u(200, 4000)), int(u(48, 300)), int(u(32, 200)), seed % 4
        kind = ("hip_fused", "hip_sparse_fused")[(seed // 4) % 2]
        densify, thr = bool((seed // 8) % 2), u(5e-6, 5e-5)
        margin = ("448,48", "320,16", "272,4", "256,0", "384,32")[(seed // 16) % 5]
        cams = fibonacci_cameras(3, W, H, seed=7100 + seed, device="cuda")
        pipe, bg, sf = PipelineParams(antialiasing=bool(seed % 2)), torch.rand(3, generator=g).cuda(), u(0.5, 3.0)
        teacher = GaussianModel.from_raw(make_gaussians(P, deg, seed=7200 + seed, scale_factor=sf).to("cuda"), requires_grad=False)
        with torch.no_grad():
            gts = {i: render(c, teacher, pipe, bg)["render"].clone() for i, c in enumerate(cams)}
        ends = []
        os.environ["GSR_CULL_MARGIN"] = margin
        try:
            for cull in (False, True):
                pool = ws.pool(torch.device("cuda", 0))
                pool.forget_estimates()
                s0 = dict(pool.stats)
                model = GaussianModel.from_raw(make_gaussians(P, deg, seed=7300 + seed, scale_factor=sf).to("cuda"))
                model.active_sh_degree = deg
                tr = Trainer(model, cams, gts, render, pipe, bg, separate_sh=True, optimizer=kind)
                if densify:
                    tr.enable_densification(extent=4.4, from_iter=2, until_iter=100, interval=7, opacity_reset_interval=50,
                                            grad_threshold=thr, min_opacity=0.005, seed=seed)
                if cull:
                    tr.enable_tile_cull()
                    tr.cull_adaptive = False       # (the sweep wants truncation tried on every frame: no back-off, no governor)
                for it in range(20):
                    tr.step(it % 3)
                tr.finish()
                torch.cuda.synchronize()
                if cull:
                    reruns += tr.rerun_views
                    s1 = dgr.call_stats()
                    culled += s1.get("culled_frames", 0) - s0.get("culled_frames", 0)
                    flagged += s1.get("cull_miss_frames", 0) - s0.get("cull_miss_frames", 0)
                st = []
                for p_ in model.parameters():
                    if p_.numel() == 0:
                        continue
                    s_ = tr.optimizer.state.get(p_, {})
                    st += [p_.detach().clone()] + ([s_["exp_avg"].clone(), s_["exp_avg_sq"].clone()] if "exp_avg" in s_ else [])
                ends.append((st + [model.xyz_gradient_accum.clone(), model.denom.clone(), model.max_radii2D.clone()], tr.iteration))
            (a, ia), (b, ib) = ends
            assert ia == ib, ("iteration", ia, ib)
            assert len(a) == len(b), ("state count", len(a), len(b))
            for i, (x, y) in enumerate(zip(a, b)):
                assert x.shape == y.shape, ("shape", i, tuple(x.shape), tuple(y.shape))
                assert torch.equal(x, y), ("value", i, float((x - y).abs().max()) if x.numel() else 0.0)
        except Exception as e:      # noqa: BLE001
            bad.append(seed)
            print(f"seed {seed} (P {P}, {W}x{H}, deg {deg}, {kind}, densify {densify}, sf {sf:.2f}, margin {margin}): "
                  f"{type(e).__name__}: {str(e)[:220]}", flush=True)
        if (seed - first) % 25 == 24:
            print(f"... {seed - first + 1} cases, {len(bad)} failures, {culled} truncated frames, {flagged} flagged, {reruns} views run "
                  f"again, {time.time() - t0:.0f} s", flush=True)
finally:
    dgr.set_forward_mode(old_mode)
    if old_margin is None:
        os.environ.pop("GSR_CULL_MARGIN", None)
    else:
        os.environ["GSR_CULL_MARGIN"] = old_margin
print(f"cull sweep: seeds {first}..{first + count - 1}: {count - len(bad)} bit-identical to the untruncated run ({culled} truncated "
      f"frames, {flagged} of them flagged and run again), {len(bad)} failed {bad}")
sys.exit(1 if bad else 0)
