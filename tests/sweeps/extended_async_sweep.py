#!/usr/bin/env python3
"""Extended sweep of the unverified forward mode under capacity trouble (GPU box, repo root):
    python tests/sweeps/extended_async_sweep.py [first] [count]
Per seed a random small training run (both folded optimizers, SH 0..3, with / without densification) in the default mode against
the same run in forward mode "async" where, at random steps, the workspace pool's capacity estimate is cut to a fraction of what
the frames need: those frames are composited from truncated lists, their backward must be a no-op on the device, the Trainer must
run them again, and the end state must equal the default mode's bit for bit."""
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
from gaussian_renderer import render, PipelineParams  # noqa: E402
from scene_utils import make_gaussians, fibonacci_cameras, GaussianModel, Trainer  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad, reruns, t0 = [], 0, time.time()
old_mode, old_min = ws.forward_mode(), ws.MIN_CAPACITY
ws.MIN_CAPACITY = 256
try:
    for seed in range(first, first + count):
        g = torch.Generator().manual_seed(4000 + seed)

        def u(a, b):
            return a + (b - a) * float(torch.rand((), generator=g))
        P, W, H, deg = int(u(50, 5000)), int(u(33, 300)), int(u(17, 200)), seed % 4
        kind = ("hip_fused", "hip_sparse_fused", "hip", "hip_sparse")[(seed // 4) % 4]
        densify, thr = bool((seed // 16) % 2), u(5e-6, 5e-5)
        cut_at = sorted({int(u(1, 11)) for _ in range(3)})
        cams = fibonacci_cameras(3, W, H, seed=4100 + seed, device="cuda")
        pipe, bg, sf = PipelineParams(), torch.rand(3, generator=g).cuda(), u(0.4, 1.5)
        teacher = GaussianModel.from_raw(make_gaussians(P, deg, seed=4200 + seed, scale_factor=sf).to("cuda"), requires_grad=False)
        with torch.no_grad():
            gts = {i: render(c, teacher, pipe, bg)["render"].clone() for i, c in enumerate(cams)}
        ends = []
        try:
            for mode in ("exact", "async"):
                dgr.set_forward_mode(mode)
                pool = ws.pool(torch.device("cuda", 0))
                pool.forget_estimates()
                model = GaussianModel.from_raw(make_gaussians(P, deg, seed=4300 + seed, scale_factor=sf).to("cuda"))
                model.active_sh_degree = deg
                tr = Trainer(model, cams, gts, render, pipe, bg, separate_sh=True, optimizer=kind)
                if densify:
                    tr.enable_densification(extent=4.4, from_iter=2, until_iter=100, interval=5, opacity_reset_interval=50,
                                            grad_threshold=thr, min_opacity=0.005, seed=seed)
                for it in range(12):
                    if mode == "async" and it in cut_at:
                        for k in list(pool.capacity):
                            pool.capacity[k] = max(256, pool.capacity[k] // 8)
                    tr.step(it % 3)
                tr.finish()
                torch.cuda.synchronize()
                reruns += tr.rerun_views
                st = []
                for p_ in model.parameters():
                    if p_.numel() == 0:
                        continue
                    s_ = tr.optimizer.state.get(p_, {})
                    st += [p_.detach().clone()] + ([s_["exp_avg"].clone(), s_["exp_avg_sq"].clone()] if "exp_avg" in s_ else [])
                ends.append((st + [model.xyz_gradient_accum.clone(), model.denom.clone(), model.max_radii2D.clone()], tr.iteration))
            (a, ia), (b, ib) = ends
            assert len(a) == len(b), ("state count", len(a), len(b))
            for i, (x, y) in enumerate(zip(a, b)):
                assert x.shape == y.shape, ("shape", i, tuple(x.shape), tuple(y.shape))
                assert torch.equal(x, y), ("value", i, float((x - y).abs().max()) if x.numel() else 0.0)
        except Exception as e:      # noqa: BLE001
            bad.append(seed)
            print(f"seed {seed} (P {P}, {W}x{H}, deg {deg}, {kind}, densify {densify}, cuts {cut_at}): {type(e).__name__}: {str(e)[:220]}", flush=True)
        if (seed - first) % 25 == 24:
            print(f"... {seed - first + 1} cases, {len(bad)} failures, {reruns} views run again, {time.time() - t0:.0f} s", flush=True)
finally:
    dgr.set_forward_mode(old_mode)
    ws.MIN_CAPACITY = old_min
print(f"async sweep: seeds {first}..{first + count - 1}: {count - len(bad)} bit-identical to the default mode ({reruns} truncated frames "
      f"run again on the way), {len(bad)} failed {bad}")
sys.exit(1 if bad else 0)
