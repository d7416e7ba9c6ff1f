"""Probe: can a forward (unverified mode) + fused loss + backward of this library be captured into a HIP graph through
torch.cuda.graph and replayed?  (GPU box, repo root.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-slam_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import diff_gaussian_rasterization as dgr  # noqa: E402
from gaussian_renderer import render, PipelineParams  # noqa: E402
from scene_utils import make_gaussians, fibonacci_cameras, GaussianModel  # noqa: E402
from fused_ssim import fused_l1_ssim_loss  # noqa: E402

dev = "cuda"
P, W, H, deg = 20000, 320, 200, 3
cams = fibonacci_cameras(2, W, H, seed=3, device=dev)
pipe, bg = PipelineParams(), torch.zeros(3, device=dev)
teacher = GaussianModel.from_raw(make_gaussians(P, deg, seed=1, scale_factor=0.7).to(dev), requires_grad=False)
with torch.no_grad():
    gt = render(cams[0], teacher, pipe, bg)["render"].clone()
model = GaussianModel.from_raw(make_gaussians(P, deg, seed=2, scale_factor=0.7).to(dev))
params = list(model.parameters())


def step():
    pkg = render(cams[0], model, pipe, bg, separate_sh=True)
    loss = fused_l1_ssim_loss(pkg["render"], gt, 0.2)
    loss.backward()
    return loss


dgr.set_forward_mode("async")
for _ in range(3):                       # warm-up: capacities, pinned slots, side streams, allocator
    for p in params:
        p.grad = None
    step()
torch.cuda.synchronize()
ref = [p.grad.clone() for p in params]
for p in params:
    p.grad.zero_()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for p in params:
        p.grad = None
    step()                               # once more on the side stream, as torch's capture recipe asks
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
for p in params:
    p.grad = None
dgr.prepare_for_graph_capture()
with torch.cuda.graph(g):
    static_loss = step()
torch.cuda.synchronize()
print("captured")
for rep in range(3):
    for p in params:
        p.grad.zero_() if False else None
    g.replay()
torch.cuda.synchronize()
print("replayed; loss", float(static_loss))
for p, r in zip(params, ref):
    print(tuple(p.shape), "equal" if torch.equal(p.grad, r) else f"DIFF {float((p.grad - r).abs().max()):.3e}")
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200):
    g.replay()
torch.cuda.synchronize(); t1 = time.perf_counter()
for _ in range(200):
    for p in params:
        p.grad = None
    step()
torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"replay {1e3 * (t1 - t0) / 200:.3f} ms/iter, eager {1e3 * (t2 - t1) / 200:.3f} ms/iter")
