"""Worker for tests/test_dp_overlap_gpu.py: launched with torch.distributed.run, 2 ranks sharing ONE GPU, gloo between them.
Trains the same scene twice - plain synchronous exchange, then the overlapped schedule - and checks the parameters are identical
on both ranks and between the two schedules."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-slam_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

os.environ["BENCH_SHARE_GPU"] = "1"
from scene_utils import init_from_env, shard_views, Trainer, GaussianModel, make_gaussians, fibonacci_cameras  # noqa: E402
from gaussian_renderer import render, PipelineParams  # noqa: E402


def run(overlap, optimizer, rank, world, steps=6, exchange="allreduce", fuse_rank1=True, reset=50):
    dev = "cuda:0"
    raw = make_gaussians(3000, 2, seed=12, scale_factor=0.8)
    cams = fibonacci_cameras(4, 128, 80, seed=13, device=dev)
    teacher = GaussianModel.from_raw(make_gaussians(3000, 2, seed=14, scale_factor=0.8).to(dev), requires_grad=False)
    bg = torch.zeros(3, device=dev)
    pipe = PipelineParams()
    with torch.no_grad():
        gts = {i: render(c, teacher, pipe, bg)["render"].clone() for i, c in enumerate(cams)}
    model = GaussianModel.from_raw(raw.to(dev))
    tr = Trainer(model, cams, gts, render, pipe, bg, world=world, rank=rank, optimizer=optimizer, separate_sh=True,
                 overlap_comm=overlap, exchange=exchange)
    assert tr.overlap_comm == overlap and tr.exchange == exchange
    tr.rank1_fuse_adam = fuse_rank1
    tr.enable_densification(extent=4.4, from_iter=2, until_iter=100, interval=4, opacity_reset_interval=reset, grad_threshold=2e-5)
    mine = shard_views(len(cams), rank, world)
    for it in range(steps):
        tr.step(mine[it % len(mine)])
    tr.finish()
    torch.cuda.synchronize()
    if exchange == "sharded":
        assert tr.sharded is not None and tr.sharded.moment_bytes() < sum(p.numel() for p in model.parameters()) * 8 * 0.6
    return [p.detach().clone() for p in model.parameters()]


def main():
    rank, world, _ = init_from_env("gloo")
    torch.cuda.set_device(0)
    for optimizer in ("hip", "hip_sparse"):
        plain = run(False, optimizer, rank, world)
        over = run(True, optimizer, rank, world)
        for a, b in zip(plain, over):
            assert a.shape == b.shape and torch.equal(a, b), f"overlapped schedule changed the result ({optimizer})"
        # both ranks hold the same parameters
        for t in over:
            other = t.clone()
            dist.broadcast(other, src=0)
            assert torch.equal(other, t), f"ranks diverged ({optimizer})"
    # exchange="sh_rank1" (SH gradients rebuilt from the all-gathered dL/df_dc): the overlapped schedule is bit-identical to the
    # plain one, both ranks agree bit for bit, and the parameters match the all-reduce schedule's to fp32 rounding
    ref = run(False, "hip", rank, world)
    plain = run(False, "hip", rank, world, exchange="sh_rank1")
    over = run(True, "hip", rank, world, exchange="sh_rank1")
    unfused = run(True, "hip", rank, world, exchange="sh_rank1", fuse_rank1=False)    # rebuild -> .grad -> optimizer.step()
    for a, b in zip(over, unfused):
        assert a.shape == b.shape and torch.equal(a, b), "the Adam step folded into the rank-one rebuild changed the result"
    for a, b, c in zip(plain, over, ref):
        assert a.shape == b.shape and torch.equal(a, b), "overlapped sh_rank1 schedule changed the result"
        assert a.shape == c.shape and torch.allclose(a, c, atol=1e-6, rtol=1e-4), float((a - c).abs().max())
        other = b.clone()
        dist.broadcast(other, src=0)
        assert torch.equal(other, b), "ranks diverged (sh_rank1)"
    # exchange="sharded" WITH densification (round 4): the moments live per row shard, a densification gathers them, carries them
    # through the row surgery and slices them again; an opacity reset replaces one tensor.  9 steps: densifications at 4 and 8, a
    # reset-only iteration at 5 (the opacity group is skipped by that step, the other five are applied) - the parameters must equal
    # the all-reduce schedule's bit for bit (Adam is element-wise), on both ranks.
    ref = run(False, "hip", rank, world, steps=9, reset=5)
    sh = run(False, "hip", rank, world, steps=9, exchange="sharded", reset=5)
    assert ref[0].shape[0] != 3000, "the scene did not densify"
    for a, b in zip(ref, sh):
        assert a.shape == b.shape and torch.equal(a, b), ("sharded + densification differs from the all-reduce schedule",
                                                          float((a - b).abs().max()))
        other = b.clone()
        dist.broadcast(other, src=0)
        assert torch.equal(other, b), "ranks diverged (sharded + densification)"
    if rank == 0:
        print("DP_OVERLAP_OK")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
