#!/usr/bin/env python3
"""Extended bit-equality sweep of the optimizer folded into the rasterizer backward (GPU box, repo root):
    python tests/sweeps/extended_fused_sweep.py [first] [count]
Per seed a random small scene (odd Gaussian counts and image sizes, SH degree 0..3, dense Adam or SparseGaussianAdam, anti-
aliasing on / off, a few views): N training steps with the optimizer as its own launch against the same steps with the update
folded into the backward's last kernel (gsr_backward_adam).  Parameters, both moments and the densification statistics must be
equal bit for bit (tests/test_loss_adam_gpu.py checks the same at two fixed sizes)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-slam_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

torch.set_num_threads(16)      # (the box's CPU share; torch's default there is 128 threads on a 16-CPU quota)

from gaussian_renderer import render, PipelineParams  # noqa: E402
from scene_utils import make_gaussians, fibonacci_cameras, GaussianModel, Trainer  # noqa: E402

DENSIFY = "--densify" in sys.argv    # ... with the reference's densify / prune / opacity-reset schedule compressed into the run
GRAPH = "--graph" in sys.argv        # second run: the step replayed from a HIP graph (Trainer.enable_graph_replay) instead of eager
args_ = [a for a in sys.argv[1:] if a not in ("--densify", "--graph")]
first = int(args_[0]) if len(args_) > 0 else 0
count = int(args_[1]) if len(args_) > 1 else 100
bad, t0 = [], time.time()
for seed in range(first, first + count):
    g = torch.Generator().manual_seed(9000 + seed)

    def u(a, b):
        return a + (b - a) * float(torch.rand((), generator=g))
    P, W, H, deg = int(u(1, 5000)), int(u(17, 300)), int(u(9, 200)), seed % 4
    kind = ("hip", "hip_sparse")[(seed // 4) % 2]
    pipe = PipelineParams()
    pipe.antialiasing = bool((seed // 8) % 2)
    cams = fibonacci_cameras(3, W, H, seed=9100 + seed, device="cuda")
    bg = torch.rand(3, generator=g).cuda()
    sf = u(0.3, 2.0)
    teacher = GaussianModel.from_raw(make_gaussians(P, deg, seed=9200 + seed, scale_factor=sf).to("cuda"), requires_grad=False)
    with torch.no_grad():
        gts = {i: render(c, teacher, pipe, bg)["render"].clone() for i, c in enumerate(cams)}
    runs = []
    thr, min_op = u(1e-6, 1e-4), u(0.001, 0.2)       # (the high end prunes a model to EMPTY behind an opacity reset: that runs too)
    try:
        for fused in (False, True):
            model = GaussianModel.from_raw(make_gaussians(P, deg, seed=9300 + seed, scale_factor=sf).to("cuda"))
            model.active_sh_degree = deg
            tr = Trainer(model, cams, gts, render, pipe, bg, separate_sh=True,
                         optimizer=kind + ("_fused" if (fused or GRAPH) else ""))
            if GRAPH and fused:
                tr.enable_graph_replay(warmup=1)
            if DENSIFY:
                tr.enable_densification(extent=4.4, from_iter=2, until_iter=100, interval=4, opacity_reset_interval=9,
                                        grad_threshold=thr, min_opacity=min_op, seed=seed)
            for it in range(14 if DENSIFY else (9 if GRAPH else 5)):
                tr.step(it % 3)
            tr.finish()
            torch.cuda.synchronize()
            st = []
            for p_ in model.parameters():
                if p_.numel() == 0:             # (SH degree 0: an empty f_rest; whether an optimizer keeps state for it is its own business)
                    continue
                s_ = tr.optimizer.state.get(p_, {})
                st += [p_.detach().clone()] + ([s_["exp_avg"].clone(), s_["exp_avg_sq"].clone()] if "exp_avg" in s_ else [])
            runs.append(st + [model.xyz_gradient_accum.clone(), model.denom.clone(), model.max_radii2D.clone()])
        assert len(runs[0]) == len(runs[1]), ("state count", len(runs[0]), len(runs[1]))
        for i, (a, b) in enumerate(zip(*runs)):
            assert a.shape == b.shape, ("shape", i, tuple(a.shape), tuple(b.shape))
            assert torch.equal(a, b), ("value", i, float((a - b).abs().max()) if a.numel() else 0.0)
    except Exception as e:      # noqa: BLE001
        bad.append(seed)
        print(f"seed {seed} (P {P}, {W}x{H}, deg {deg}, {kind}, aa {pipe.antialiasing}): {type(e).__name__}: {str(e)[:200]}", flush=True)
    if (seed - first) % 25 == 24:
        print(f"... {seed - first + 1} cases, {len(bad)} failures, {time.time() - t0:.0f} s", flush=True)
print(f"fused-optimizer sweep: seeds {first}..{first + count - 1}: {count - len(bad)} bit-identical, {len(bad)} failed {bad}")
sys.exit(1 if bad else 0)
