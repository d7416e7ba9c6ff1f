"""Worker for tests/test_dp_overlap_gpu.py::test_every_exchange_runs_on_rccl_with_one_rank: ONE process, a process group of one
rank over RCCL (backend "nccl"), the Trainer told to run the N > 1 schedule on it (single_rank_group).  The mean over one rank
is the rank's own gradient, so every exchange form must reproduce the plain single-GPU training run: bit for bit where the
schedule only moves data (all-reduce AVG, reduce-scatter + all-gather, visible rows), to fp32 rounding for sh_rank1 (which
rebuilds the SH gradients from dL/df_dc)."""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-slam_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from scene_utils import Trainer, GaussianModel, make_gaussians, fibonacci_cameras  # noqa: E402
from gaussian_renderer import render, PipelineParams  # noqa: E402


def run(group, exchange="allreduce", overlap=False, optimizer="hip", densify=True, steps=6):
    dev = "cuda:0"
    raw = make_gaussians(3000, 2, seed=12, scale_factor=0.8)
    cams = fibonacci_cameras(4, 128, 80, seed=13, device=dev)
    teacher = GaussianModel.from_raw(make_gaussians(3000, 2, seed=14, scale_factor=0.8).to(dev), requires_grad=False)
    bg = torch.zeros(3, device=dev)
    pipe = PipelineParams()
    with torch.no_grad():
        gts = {i: render(c, teacher, pipe, bg)["render"].clone() for i, c in enumerate(cams)}
    model = GaussianModel.from_raw(raw.to(dev))
    tr = Trainer(model, cams, gts, render, pipe, bg, world=1, rank=0, optimizer=optimizer, separate_sh=True,
                 overlap_comm=overlap, exchange=exchange, single_rank_group=group)
    assert tr.distributed == group and tr.overlap_comm == (overlap and group)
    if densify:
        tr.enable_densification(extent=4.4, from_iter=2, until_iter=100, interval=4, opacity_reset_interval=50,
                                grad_threshold=2e-5)
    for it in range(steps):
        tr.step(it % len(cams))
    tr.finish()
    torch.cuda.synchronize()
    return [p.detach().clone() for p in model.parameters()]


def same(a, b, what):
    for x, y in zip(a, b):
        assert x.shape == y.shape and torch.equal(x, y), what


def main():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    assert dist.get_backend() == "nccl"
    for optimizer in ("hip", "hip_sparse"):
        ref = run(False, optimizer=optimizer)
        same(ref, run(True, "allreduce", False, optimizer), f"all-reduce over RCCL changed the result ({optimizer})")
        same(ref, run(True, "allreduce", True, optimizer), f"overlapped all-reduce over RCCL changed the result ({optimizer})")
        same(ref, run(True, "visible_rows", False, optimizer), f"visible-rows exchange over RCCL changed the result ({optimizer})")
    ref = run(False, densify=False)
    same(ref, run(True, "sharded", False, densify=False), "reduce-scatter / all-gather over RCCL changed the result")
    ref = run(False)
    plain, over = run(True, "sh_rank1", False), run(True, "sh_rank1", True)
    same(plain, over, "overlapped sh_rank1 over RCCL changed the result")
    for a, c in zip(plain, ref):
        assert a.shape == c.shape and torch.allclose(a, c, atol=1e-6, rtol=1e-4), float((a - c).abs().max())
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_SINGLE_RANK_OK")


if __name__ == "__main__":
    main()
