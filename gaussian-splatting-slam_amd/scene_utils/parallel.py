"""View-sharded data parallelism for the rasterizer path.

The reference is single-GPU (utils/general_utils.py:133 pins cuda:0; no collective anywhere).  Views are
independent units: every rank holds the full Gaussian set, renders views {v : v mod world == rank} and the six
leaf gradients (59 floats per Gaussian at SH degree 3) are averaged in place by back-to-back all-reduces per step
(RCCL over xGMI when the backend is "nccl").  Densification statistics are
per-view quantities and are reduced separately (`reduce_densification_stats`).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None):
    """-> (rank, world, local_rank).  No-op (0,1,0) when not launched by torchrun."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1, 0
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl" and not os.environ.get("BENCH_SHARE_GPU"):
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_views(n_views: int, rank: int, world: int, epoch_perm=None):
    """View ids rendered by `rank` (round-robin over a shared permutation so every rank sees a different view
    at every step and all views are covered exactly once per epoch)."""
    ids = list(range(n_views)) if epoch_perm is None else list(epoch_perm)
    return ids[rank::world]


class GradBucket:
    """Sums the gradients of `params` over ranks and leaves the MEAN in `.grad`.

    The six leaf gradients are reduced IN PLACE, one collective each, issued back-to-back as async work (RCCL runs
    them in order on its own stream; the small ones pipeline behind the 180 MB SH-rest tensor).  Flattening them into
    one buffer would add a 236 MB copy-in and copy-out per step (~0.25 ms on MI355X, ~7 % of the step) for nothing:
    the collective is bandwidth-bound on the xGMI links either way."""

    def __init__(self, params):
        self.params = list(params)

    def all_reduce_mean(self, world: int, group=None, params=None):
        """`params`: optional subset of the bucket's parameters to exchange in this call."""
        if world <= 1:
            return
        backend = dist.get_backend(group)
        use_avg = backend == "nccl"                  # RCCL averages in the reduction; gloo has no AVG
        todo = self.params if params is None else list(params)
        works = []
        for p in todo:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            g = p.grad
            if not g.is_contiguous():
                g = p.grad = g.contiguous()
            works.append(dist.all_reduce(g, op=dist.ReduceOp.AVG if use_avg else dist.ReduceOp.SUM, group=group,
                                         async_op=True))
        for w in works:
            w.wait()
        if not use_avg:
            for p in todo:
                p.grad.mul_(1.0 / world)


def reduce_densification_stats(xyz_gradient_accum, denom, max_radii2D, world: int, group=None):
    """Per-view statistics (reference scene/gaussian_model.py:431-433, train.py:159) -> identical on all ranks."""
    if world <= 1:
        return
    dist.all_reduce(xyz_gradient_accum, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(denom, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX, group=group)
