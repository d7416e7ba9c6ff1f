"""View-sharded data parallelism for the rasterizer path.

The reference is single-GPU (utils/general_utils.py:133 pins cuda:0; no collective anywhere).  Views are
independent units: every rank holds the full Gaussian set, renders views {v : v mod world == rank} and the six
leaf gradients (59 floats per Gaussian at SH degree 3) are summed with ONE flattened all-reduce per step
(RCCL over xGMI when the backend is "nccl"), then divided by the world size.  Densification statistics are
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
        if backend == "nccl":
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
    """Flattens the gradients of `params` into one contiguous fp32 buffer for a single all-reduce."""

    def __init__(self, params):
        self.params = list(params)
        self.numel = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=p0.device)
        self.views = []
        o = 0
        for p in self.params:
            self.views.append(self.flat[o:o + p.numel()].view_as(p))
            o += p.numel()

    def resize_like_params(self):
        if sum(p.numel() for p in self.params) != self.numel:
            self.__init__(self.params)

    def all_reduce_mean(self, world: int, group=None):
        """Sums the params' .grad over ranks and writes the mean back into .grad."""
        if world <= 1:
            return
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        self.flat.mul_(1.0 / world)
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                p.grad = v.clone()
            else:
                p.grad.copy_(v)


def reduce_densification_stats(xyz_gradient_accum, denom, max_radii2D, world: int, group=None):
    """Per-view statistics (reference scene/gaussian_model.py:431-433, train.py:159) -> identical on all ranks."""
    if world <= 1:
        return
    dist.all_reduce(xyz_gradient_accum, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(denom, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX, group=group)
