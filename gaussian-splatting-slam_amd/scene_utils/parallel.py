"""View-sharded data parallelism for the rasterizer path.

The reference is single-GPU (utils/general_utils.py:133 pins cuda:0; no collective anywhere).  Views are
independent units: every rank holds the full Gaussian set, renders views {v : v mod world == rank} and the six
leaf gradients (59 floats per Gaussian at SH degree 3) are averaged in place, as ONE flat span per exchange
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


ARENA_ALIGN = 64      # floats; = diff_gaussian_rasterization.GRAD_ARENA_ALIGN (asserted by tests/test_distributed_cpu.py)
ARENA_SPARE = 3       # floats kept free behind every tensor of an arena (the sh_rank1 exchange's camera-centre row)


def canonical_offsets(numels):
    """Element offsets of tensors with `numels` elements laid out side by side the way the rasterizer's backward lays out its
    gradient arena (diff_gaussian_rasterization._grad_arena): in list order, ARENA_SPARE floats free behind each, each start on
    an ARENA_ALIGN boundary.  -> (offsets, total span length = last offset + last numel).  A function of the SIZES only, so
    every rank computes the same plan."""
    offs, cur = [], 0
    for n in numels:
        offs.append(cur)
        cur += -(-(int(n) + ARENA_SPARE) // ARENA_ALIGN) * ARENA_ALIGN
    return offs, (offs[-1] + int(numels[-1]) if offs else 0)


def flat_span(tensors):
    """ONE flat tensor (the tensors' common dtype) of canonical_offsets' length over `tensors` (in list order).  If they already sit at exactly those
    offsets in one storage (the rasterizer's gradient arena does that), the span is a view of it - (flat, None); otherwise a
    zero-filled scratch buffer with the tensors copied in - (flat, [views to copy back from]).  Either way the collective that
    travels is the same size on every rank: the plan never depends on a rank's local memory layout."""
    offs, total = canonical_offsets([t.numel() for t in tensors])
    t0 = tensors[0]
    base = t0.storage_offset()
    same = all(t.dtype == t0.dtype and t.is_contiguous() and t.device == t0.device and
               t.untyped_storage().data_ptr() == t0.untyped_storage().data_ptr() and t.storage_offset() - base == o
               for t, o in zip(tensors, offs))
    if same and (base + total) * t0.element_size() <= t0.untyped_storage().nbytes():
        flat = torch.empty(0, dtype=t0.dtype, device=t0.device).set_(t0.untyped_storage(), base, (total,), (1,))
        return flat, None
    flat = torch.zeros(total, dtype=t0.dtype, device=t0.device)
    views = [flat[o:o + t.numel()].view(t.shape) for t, o in zip(tensors, offs)]
    for v, t in zip(views, tensors):
        v.copy_(t)
    return flat, views


class GradBucket:
    """Sums the gradients of `params` over ranks and leaves the MEAN in `.grad`.

    One call = ONE collective over one flat span that holds the gradients side by side in the canonical arena layout
    (`canonical_offsets`: a function of the tensor sizes and their order only, hence identical on every rank).  The rasterizer's
    backward returns its gradients in exactly that layout (one allocation, geometry first), so on the training path the span
    is a view and nothing is copied; a rank whose gradients live elsewhere (a `.grad` autograd cloned, zeros for a missing
    gradient, gradients that came through torch activations) copies them into a scratch span and back - a local cost that never
    changes the number or the size of the collectives its peers see.  Pass the parameters in the arena's order (xyz, opacity,
    scaling, rotation, f_dc, f_rest) for the zero-copy path.  (The ARENA_SPARE / alignment floats between the tensors travel
    too; nothing reads them.)"""

    def __init__(self, params):
        self.params = list(params)
        self.copied_spans = 0        # calls whose span had to be assembled by copies (diagnostic)

    def all_reduce_mean(self, world: int, group=None, params=None, visible=None, force=False):
        """`params`: optional subset of the bucket's parameters to exchange in this call (in the arena's order).
        `visible` (bool[P], this rank's visibility filter): exchange only the rows some rank saw (`all_reduce_visible_rows`).
        `force`: run the collectives although world == 1 (the one-rank RCCL rehearsal, Trainer(single_rank_group=True))."""
        if world <= 1 and not force:
            return
        if visible is not None:
            return self.all_reduce_visible_rows(world, visible, group=group, params=params)
        backend = dist.get_backend(group)
        use_avg = backend == "nccl"                  # RCCL averages in the reduction; gloo has no AVG
        todo = [p for p in (self.params if params is None else list(params)) if p.numel() > 0]
        if not todo:
            return
        for p in todo:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            if not p.grad.is_contiguous() or p.grad.dtype != p.dtype:
                p.grad = p.grad.contiguous().to(p.dtype)
        flat, views = flat_span([p.grad for p in todo])
        dist.all_reduce(flat, op=dist.ReduceOp.AVG if use_avg else dist.ReduceOp.SUM, group=group)
        if not use_avg:
            flat.mul_(1.0 / world)
        if views is not None:
            self.copied_spans += 1
            for p, v in zip(todo, views):
                p.grad.copy_(v)

    def all_reduce_visible_rows(self, world: int, visible, group=None, params=None):
        """Visible-rows-only exchange: a Gaussian culled on EVERY rank has an exactly zero gradient everywhere (the rasterizer
        writes zeros for rows without instances), so only the union of the ranks' visible rows needs to travel: one 1-byte-per-
        row MAX all-reduce of the masks, then the six tensors are exchanged as compacted [V, row] buffers and scattered back.
        Same result as `all_reduce_mean`, bit for bit; pays off when V << P (room-scale captures where a view sees a fraction
        of the map), costs one host read of V (`nonzero`) per step."""
        todo = self.params if params is None else list(params)
        u8 = visible.to(torch.uint8)
        dist.all_reduce(u8, op=dist.ReduceOp.MAX, group=group)
        idx = u8.nonzero(as_tuple=True)[0]
        use_avg = dist.get_backend(group) == "nccl"
        bufs, works = [], []
        for p in todo:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            buf = p.grad.index_select(0, idx)
            bufs.append(buf)
            works.append(dist.all_reduce(buf, op=dist.ReduceOp.AVG if use_avg else dist.ReduceOp.SUM, group=group,
                                         async_op=True))
        for w, p, buf in zip(works, todo, bufs):
            w.wait()
            if not use_avg:
                buf.mul_(1.0 / world)
            p.grad.index_copy_(0, idx, buf)
        return int(idx.numel())


def _fused_sh_adam_args(optimizer, f_dc, f_rest):
    """gsr_fused_adam for the f_dc / f_rest groups of a dense FusedAdam (their step counters advance), or None."""
    from diff_gaussian_rasterization import _C, FusedAdam
    if not isinstance(optimizer, FusedAdam):
        return None
    groups = {g.get("name"): g for g in optimizer.param_groups}
    if "f_dc" not in groups or "f_rest" not in groups or groups["f_dc"]["params"][0] is not f_dc or \
            groups["f_rest"]["params"][0] is not f_rest:
        return None
    fa = _C.gsr_fused_adam()
    keep = []
    for i, (name, p) in ((1, ("f_dc", f_dc)), (2, ("f_rest", f_rest))):
        if not p.is_contiguous() or p.dtype != torch.float32:
            return None
        st = optimizer.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        keep.append(st)
    for i, (name, p) in ((1, ("f_dc", f_dc)), (2, ("f_rest", f_rest))):
        st = optimizer.state[p]
        st["step"] += 1
        fa.exp_avg[i] = st["exp_avg"].data_ptr() if p.numel() else None
        fa.exp_avg_sq[i] = st["exp_avg_sq"].data_ptr() if p.numel() else None
        fa.lr[i] = float(groups[name]["lr"])
        fa.step[i] = int(st["step"])
    g0 = optimizer.param_groups[0]
    fa.beta1, fa.beta2, fa.eps = float(g0["betas"][0]), float(g0["betas"][1]), float(g0["eps"])
    fa.sparse = 0
    return fa, keep


def rank1_sh_exchange(xyz, f_dc, f_rest, cam_center, sh_degree: int, world: int, group=None, optimizer=None):
    """Exchange of the two SH gradient tensors for ONE view per rank per step (`exchange="sh_rank1"`).

    A view's SH gradient is rank one per Gaussian: dL/dsh[k][c] = basis_k(dir) * dL/drgb_c (SURVEY.md A.7 iv), and band 0's
    basis is the constant C0, so this rank's `f_dc.grad` (= C0 * dL/drgb, zero where the colour was clamped) holds all the
    information `f_rest.grad` [P, 15, 3] is made of.  The ranks all-gather f_dc.grad and their camera centre - 12 B per Gaussian
    and rank instead of all-reducing 192 B - and every rank rebuilds the MEAN of both gradients locally, summing the ranks in
    order (csrc/exchange.hip on the HIP device; the same arithmetic in torch ops on the CPU for the gloo tests): identical bits
    on every rank; against the all-reduce schedule the values agree to fp32 rounding of the individual products
    ((b / C0) (C0 g) instead of b g, and a fixed instead of the collective's summation order).
    `xyz`: the positions the forwards saw (call before the optimizer moves them).  Leaves the means in f_dc.grad / f_rest.grad -
    or, with `optimizer` (a dense FusedAdam whose f_dc / f_rest groups are these tensors, HIP device), applies that optimizer's
    step to the two tensors in the rebuilding kernel itself (gsr_sh_rank1_adam: the rebuilt 192 B per Gaussian are never
    written; bit-identical to rebuild + optimizer.step(only=("f_dc", "f_rest"))), clears both .grad and returns True."""
    P = int(xyz.shape[0])
    krest = int(f_rest.shape[1]) if f_rest.numel() else 0
    g = f_dc.grad if f_dc.grad is not None else torch.zeros_like(f_dc)
    mine = None
    if g.is_cuda:
        from diff_gaussian_rasterization import dc_grad_tail_row
        mine = dc_grad_tail_row(g)                   # the backward left a spare row behind its dc gradient: no concatenation
        if mine is not None:
            mine[P].copy_(cam_center.reshape(3).to(g))
    if mine is None:
        mine = torch.cat((g.reshape(P, 3), cam_center.reshape(1, 3).to(g)), dim=0).contiguous()    # [P + 1, 3]
    gathered = torch.empty(world * (P + 1), 3, dtype=g.dtype, device=g.device)
    dist.all_gather_into_tensor(gathered, mine, group=group)
    if g.is_cuda and optimizer is not None:
        fused = _fused_sh_adam_args(optimizer, f_dc, f_rest)
        if fused is not None:
            import ctypes as C
            from diff_gaussian_rasterization import _C
            with _C.on_device(g.device):
                _C.check(_C.lib().gsr_sh_rank1_adam(P, world, int(sh_degree), krest, _C.ptr(xyz.detach().contiguous()),
                                                    _C.ptr(gathered), C.c_float(1.0 / world), _C.ptr(f_dc.data),
                                                    _C.ptr(f_rest.data) if krest else None, C.byref(fused[0]), _C._stream()))
            f_dc.grad = None
            f_rest.grad = None
            return True
    out_dc = torch.empty_like(f_dc)
    out_rest = torch.empty_like(f_rest)
    if g.is_cuda:
        import ctypes as C
        from diff_gaussian_rasterization import _C
        with _C.on_device(g.device):
            _C.check(_C.lib().gsr_sh_rank1_expand(P, world, int(sh_degree), krest, _C.ptr(xyz.detach().contiguous()),
                                                  _C.ptr(gathered), C.c_float(1.0 / world), _C.ptr(out_dc),
                                                  _C.ptr(out_rest) if krest else None, _C._stream()))
    else:
        from .sh import sh_basis, C0
        gathered = gathered.view(world, P + 1, 3)
        acc_dc = torch.zeros(P, 3, dtype=g.dtype)
        acc = torch.zeros(P, krest, 3, dtype=g.dtype)
        K = (sh_degree + 1) ** 2
        for r in range(world):
            gr, cam = gathered[r, :P], gathered[r, P]
            acc_dc += gr
            if krest and K > 1:
                d = xyz.detach() - cam
                d = d / d.norm(dim=1, keepdim=True)
                w = sh_basis(sh_degree, d)[:, 1:K] * (1.0 / C0)                                  # [P, K - 1]
                acc[:, :K - 1] += w[:, :, None] * gr[:, None, :]
        out_dc.copy_((acc_dc * (1.0 / world)).view_as(out_dc))
        out_rest.copy_(acc * (1.0 / world))
    f_dc.grad = out_dc
    f_rest.grad = out_rest
    return False


def exchange_bytes_per_gaussian(exchange: str, world: int, sh_coeffs: int = 16, visible_fraction: float = 1.0):
    """Bytes a rank RECEIVES per Gaussian and step over the links (ring / direct algorithms: an all-reduce of b bytes moves
    2 (N-1)/N b, reduce-scatter and all-gather (N-1)/N b each, an all-gather of b bytes per rank (N-1) b) - the table of DESIGN.md 5."""
    n = world
    row = 4 * (11 + 3 * sh_coeffs)                     # 59 floats at SH degree 3
    f = (n - 1) / n
    if exchange == "allreduce":
        return 2 * f * row
    if exchange == "sharded":
        return f * row + f * row
    if exchange == "visible_rows":
        return 2 * f * (1 + row * visible_fraction)    # + the 1-byte mask
    if exchange == "sh_rank1":
        return 2 * f * 4 * 11 + (n - 1) * 12           # geometry all-reduced, dL/df_dc all-gathered
    raise ValueError(exchange)


class ShardedStep:
    """reduce-scatter -> optimizer on a 1/N ROW shard -> all-gather of the parameters (the exchange DESIGN.md 5 sizes for xGMI).

    The all-reduce schedule moves 2 (N-1)/N x 236 B per Gaussian per rank AND lets every rank run the full Adam update (1.65 GB
    of HBM traffic at 1 M Gaussians) on identical data.  Here rank r owns rows [r c, (r+1) c), c = P // N: the gradients are
    reduce-scattered ((N-1)/N x 236 B per Gaussian in), the rank updates ITS rows only - Adam traffic and both moments shrink
    to 1/N: the moments exist only for the shard - and the updated rows are all-gathered ((N-1)/N x 236 B out): the same bytes
    on the links as the all-reduce, 1/N of the optimizer work.  The P mod N rows left over are all-reduced and updated by every
    rank (identical inputs, identical results).  Element-wise optimizers only (Adam): a row's update does not depend on other
    rows, so the parameters equal those of the all-reduce schedule bit for bit."""

    def __init__(self, model, make_optimizer, world: int, rank: int, group=None, init_from=None):
        """init_from: an optimizer over the model's FULL parameters whose state (step, exp_avg, exp_avg_sq for every row) this
        rank's shard starts from - how the moments come back after a densification re-shaped the model (`full_moments`)."""
        self.model, self.world, self.rank, self.group = model, world, rank, group
        P = int(model.get_xyz.shape[0])
        self.c = P // world
        self.P0 = self.c * world
        self.items, groups = [], []
        for g in model.param_groups():
            p = g["params"][0]
            shard = torch.nn.Parameter(p.data[rank * self.c:(rank + 1) * self.c])          # shares p's storage
            rest = torch.nn.Parameter(p.data[self.P0:]) if P > self.P0 else None
            self.items.append((p, shard, rest))
            groups.append({"params": [shard] + ([rest] if rest is not None else []), "lr": g["lr"], "name": g["name"]})
        self.optimizer = make_optimizer(groups)
        if init_from is not None:
            for p, shard, rest in self.items:
                st = init_from.state.get(p)
                if not st or "exp_avg" not in st:
                    continue
                for part, rows in ((shard, slice(rank * self.c, (rank + 1) * self.c)), (rest, slice(self.P0, None))):
                    if part is None or (part is shard and self.c == 0):
                        continue
                    self.optimizer.state[part] = {"step": st["step"].clone(), "exp_avg": st["exp_avg"][rows].clone(),
                                                  "exp_avg_sq": st["exp_avg_sq"][rows].clone()}

    @torch.no_grad()
    def full_moments(self):
        """Per parameter (exp_avg, exp_avg_sq, step) for ALL rows - the ranks' shard moments all-gathered, the left-over rows'
        (identical on every rank) appended - or None where no step has created them yet.  What a densification needs: its row
        surgery (clone / split / prune) acts on whole tensors; ShardedStep(init_from=...) slices the result again."""
        out = []
        for p, shard, rest in self.items:
            st = self.optimizer.state.get(shard) if self.c > 0 else (self.optimizer.state.get(rest) if rest is not None else None)
            if not st or "exp_avg" not in st:
                out.append(None)
                continue
            m, v = torch.empty_like(p.data), torch.empty_like(p.data)
            if self.c > 0:
                dist.all_gather_into_tensor(m[:self.P0], st["exp_avg"].contiguous(), group=self.group)
                dist.all_gather_into_tensor(v[:self.P0], st["exp_avg_sq"].contiguous(), group=self.group)
            if rest is not None:
                sr = self.optimizer.state[rest]
                m[self.P0:] = sr["exp_avg"]
                v[self.P0:] = sr["exp_avg_sq"]
            out.append((m, v, st["step"]))
        return out

    @torch.no_grad()
    def step(self, skip=()):
        """skip: parameters left alone this step - no exchange, no update (an opacity tensor that `reset_opacity` is about to
        replace: the reference's optimizer skips it on that iteration because the new tensor carries no gradient)."""
        world, rank, c, P0 = self.world, self.rank, self.c, self.P0
        nccl = dist.get_backend(self.group) == "nccl"
        works, scaled = [], []
        for p, shard, rest in self.items:
            if any(p is q for q in skip):
                shard.grad = None
                if rest is not None:
                    rest.grad = None
                continue
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            g = g.contiguous()
            if c > 0:
                if nccl:
                    # (SUM, scaled below: RCCL 2.26's ONE-rank reduce-scatter with AVG leaves the last element of some
                    # sizes unwritten - 5745 floats, found by tests/sweeps/extended_exchange_sweep.py; SUM is the well-trodden path)
                    out = torch.empty_like(shard)
                    works.append(dist.reduce_scatter_tensor(out, g[:P0], op=dist.ReduceOp.SUM, group=self.group,
                                                            async_op=True))
                    shard.grad = out
                    scaled.append(out)
                else:       # gloo has no reduce-scatter: all-reduce, keep the own rows (CPU rehearsal of the same arithmetic)
                    dist.all_reduce(g[:P0], op=dist.ReduceOp.SUM, group=self.group)
                    shard.grad = g[rank * c:(rank + 1) * c] * (1.0 / world)
            if rest is not None:
                if nccl:
                    works.append(dist.all_reduce(g[P0:], op=dist.ReduceOp.AVG, group=self.group, async_op=True))
                    rest.grad = g[P0:]
                else:
                    dist.all_reduce(g[P0:], op=dist.ReduceOp.SUM, group=self.group)
                    rest.grad = g[P0:] * (1.0 / world)
        for w in works:
            w.wait()
        for out in scaled:
            out.mul_(1.0 / world)
        self.optimizer.step()
        self.optimizer.zero_grad(set_to_none=True)
        works = []
        for p, shard, rest in self.items:
            if any(p is q for q in skip):
                p.grad = None
                continue
            if c > 0:
                # (RCCL gathers in place when the input is the output's own slice; gloo gets a copy of the shard)
                src = shard.data if nccl else shard.data.clone()
                works.append(dist.all_gather_into_tensor(p.data[:P0], src, group=self.group, async_op=True))
            p.grad = None
        for w in works:
            w.wait()

    def moment_bytes(self):
        return sum(t.numel() * 4 for st in self.optimizer.state.values() for k, t in st.items()
                   if k in ("exp_avg", "exp_avg_sq"))


def reduce_densification_stats(xyz_gradient_accum, denom, max_radii2D, world: int, group=None, force=False):
    """Per-view statistics (reference scene/gaussian_model.py:431-433, train.py:159) -> identical on all ranks.
    world <= 1: nothing to do, whatever process groups exist (independent replicas under one launcher stay independent);
    `force` runs the collectives anyway (the one-rank rehearsal)."""
    if world <= 1 and not force:
        return
    dist.all_reduce(xyz_gradient_accum, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(denom, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX, group=group)
