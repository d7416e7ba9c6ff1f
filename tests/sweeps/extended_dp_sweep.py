#!/usr/bin/env python3
"""Extended sweep of the two-rank data-parallel schedules (GPU box, repo root; two ranks share the box's one GPU, gloo between them):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 \\
        tests/sweeps/extended_dp_sweep.py [first] [count]
Per seed a random small scene: for both optimizers the plain synchronous exchange against the side-stream overlap (bit for bit,
and both ranks identical), `visible_rows` and `sharded` against `allreduce` (bit for bit), `sh_rank1` plain against overlapped
against un-fused (bit for bit) and against `allreduce` (fp32 rounding) - tests/dp_overlap_worker.py on random shapes."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-slam_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

torch.set_num_threads(16)      # (the box's CPU share; torch's default there is 128 threads on a 16-CPU quota)
import torch.distributed as dist  # noqa: E402

os.environ["BENCH_SHARE_GPU"] = "1"
from scene_utils import init_from_env, shard_views, Trainer, GaussianModel, make_gaussians, fibonacci_cameras  # noqa: E402
from gaussian_renderer import render, PipelineParams  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rank, world, _ = init_from_env("gloo")
torch.cuda.set_device(0)
dev = "cuda:0"
bad, t0 = [], time.time()
for seed in range(first, first + count):
    g = torch.Generator().manual_seed(3000 + seed)

    def u(a, b):
        return a + (b - a) * float(torch.rand((), generator=g))
    P, W, H, deg = int(u(3, 4000)), int(u(17, 220)), int(u(9, 150)), seed % 4
    densify, thr, sf = bool((seed // 4) % 2), u(5e-6, 5e-5), u(0.4, 1.5)
    cams = fibonacci_cameras(4, W, H, seed=3100 + seed, device=dev)
    pipe, bg = PipelineParams(), torch.rand(3, generator=g).to(dev)
    teacher = GaussianModel.from_raw(make_gaussians(P, deg, seed=3200 + seed, scale_factor=sf).to(dev), requires_grad=False)
    with torch.no_grad():
        gts = {i: render(c, teacher, pipe, bg)["render"].clone() for i, c in enumerate(cams)}

    def run(overlap, optimizer, exchange="allreduce", fuse_rank1=True, dens=densify):
        model = GaussianModel.from_raw(make_gaussians(P, deg, seed=3300 + seed, scale_factor=sf).to(dev))
        model.active_sh_degree = deg
        tr = Trainer(model, cams, gts, render, pipe, bg, world=world, rank=rank, optimizer=optimizer, separate_sh=True,
                     overlap_comm=overlap, exchange=exchange)
        tr.rank1_fuse_adam = fuse_rank1
        if dens:
            tr.enable_densification(extent=4.4, from_iter=2, until_iter=100, interval=4, opacity_reset_interval=50,
                                    grad_threshold=thr, min_opacity=0.005, seed=seed)
        mine = shard_views(len(cams), rank, world)
        for it in range(7):
            tr.step(mine[it % len(mine)])
        tr.finish()
        torch.cuda.synchronize()
        return [p.detach().clone() for p in model.parameters()]

    def same(a, b, what, exact=True):
        for i, (x, y) in enumerate(zip(a, b)):
            assert x.shape == y.shape, (what, "shape", i)
            if exact:
                assert torch.equal(x, y), (what, i, float((x - y).abs().max()) if x.numel() else 0.0)
            else:
                assert torch.allclose(x, y, atol=2e-6, rtol=2e-4), (what, i, float((x - y).abs().max()) if x.numel() else 0.0)

    def ranks_agree(a, what):
        for t in a:
            other = t.clone()
            dist.broadcast(other, src=0)
            assert torch.equal(other, t), (what, "ranks diverged")
    ok = 1
    try:
        for optimizer in ("hip", "hip_sparse"):
            plain = run(False, optimizer)
            over = run(True, optimizer)
            same(plain, over, f"overlap ({optimizer})")
            ranks_agree(over, f"overlap ({optimizer})")
            same(plain, run(False, optimizer, "visible_rows"), f"visible_rows ({optimizer})")
        ref = run(False, "hip", dens=False)
        same(ref, run(False, "hip", "sharded", dens=False), "sharded")
        r1 = run(False, "hip", "sh_rank1")
        r1o = run(True, "hip", "sh_rank1")
        same(r1, r1o, "sh_rank1 overlap")
        same(r1o, run(True, "hip", "sh_rank1", fuse_rank1=False), "sh_rank1 un-fused")
        ranks_agree(r1o, "sh_rank1")
        if not densify:
            same(run(False, "hip"), r1, "sh_rank1 vs allreduce", exact=False)
    except Exception as e:      # noqa: BLE001
        ok = 0
        print(f"[rank {rank}] seed {seed} (P {P}, {W}x{H}, deg {deg}, densify {densify}): {type(e).__name__}: {str(e)[:200]}", flush=True)
    flag = torch.tensor([ok], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)          # (a rank that failed must not leave the other one inside a collective)
    if int(flag) == 0:
        bad.append(seed)
        if ok:       # the OTHER rank failed mid-schedule: this one may hang in its next collective - stop here
            pass
        break
    if rank == 0 and (seed - first) % 10 == 9:
        print(f"... {seed - first + 1} cases, {len(bad)} failures, {time.time() - t0:.0f} s", flush=True)
if rank == 0:
    print(f"two-rank sweep: seeds {first}..{first + count - 1}: " + ("all passed" if not bad else f"stopped at failing seed {bad}"))
dist.barrier()
dist.destroy_process_group()
sys.exit(1 if bad else 0)
