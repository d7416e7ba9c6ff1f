"""world_size-2 gloo tests of the view-sharded data-parallel path on CPU.  The renderer injected into
the Trainer here is the CPU oracle (tests may use it); the sharding / bucketed all-reduce / statistics logic is the
product code that bench.py runs over RCCL."""
import math
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_render(cam, pc, pipe, bg, separate_sh=False, **kw):
    from oracle import gs_oracle as O
    m2d = torch.zeros_like(pc.get_xyz, requires_grad=True) + 0
    m2d.retain_grad()
    s = O.OracleSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg,
                         1.0, cam.world_view_transform, cam.full_proj_transform, pc.active_sh_degree,
                         cam.camera_center, False, False, False)
    color, radii, invd = O.rasterize(pc.get_xyz, m2d, pc.get_opacity, s, shs=pc.get_features,
                                     scales=pc.get_scaling, rotations=pc.get_rotation)
    return {"render": color, "viewspace_points": m2d, "visibility_filter": radii > 0, "radii": radii, "depth": invd}


def _scene():
    from scene_utils import make_gaussians, fibonacci_cameras, GaussianModel
    raw = make_gaussians(120, 1, seed=5, scale_factor=1.2)
    cams = fibonacci_cameras(4, 32, 32, seed=6)
    gts = {i: torch.rand(3, 32, 32, generator=torch.Generator().manual_seed(100 + i)) for i in range(4)}
    return raw, cams, gts


def _worker(rank, world, port, out_dir, exchange="allreduce", steps=1):
    for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-slam_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from scene_utils import init_from_env, shard_views, Trainer, GaussianModel, reduce_densification_stats
    from oracle.loss_oracle import training_loss
    from gaussian_renderer import PipelineParams
    r, w, _ = init_from_env("gloo")
    assert (r, w) == (rank, world)
    raw, cams, gts = _scene()
    model = GaussianModel.from_raw(raw)
    tr = Trainer(model, cams, gts, _oracle_render, PipelineParams(), torch.zeros(3), world=w, rank=r, optimizer="torch", loss=training_loss,
                 exchange=exchange)
    mine = shard_views(len(cams), r, w)
    for v in mine[:steps]:
        tr.step(v)
    reduce_densification_stats(tr.xyz_gradient_accum, tr.denom, tr.max_radii2D, w)
    torch.save({"params": [p.detach().clone() for p in model.parameters()], "accum": tr.xyz_gradient_accum,
                "denom": tr.denom, "maxr": tr.max_radii2D, "views": mine,
                "moment_bytes": tr.sharded.moment_bytes() if tr.sharded is not None else None},
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_views_partition():
    from scene_utils import shard_views
    for world in (1, 2, 3, 8):
        parts = [shard_views(100, r, world) for r in range(world)]
        flat = sorted(v for p in parts for v in p)
        assert flat == list(range(100))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


@pytest.mark.timeout(600)
def test_two_rank_step_equals_mean_gradient_step(tmp_path):
    """After one DP step both ranks hold identical parameters, equal to a single-process Adam step on the MEAN of the two
    per-view gradients; densification statistics are the SUM / MAX over views."""
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(os.path.join(tmp_path, "r0.pt"))
    b = torch.load(os.path.join(tmp_path, "r1.pt"))
    assert a["views"] == [0, 2] and b["views"] == [1, 3]
    for pa, pb in zip(a["params"], b["params"]):
        assert torch.equal(pa, pb)
    assert torch.equal(a["accum"], b["accum"]) and torch.equal(a["denom"], b["denom"]) and torch.equal(a["maxr"], b["maxr"])

    # single-process reference of the same step
    sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-slam_amd"))
    from scene_utils import GaussianModel
    from oracle.loss_oracle import training_loss
    from gaussian_renderer import PipelineParams
    raw, cams, gts = _scene()
    model = GaussianModel.from_raw(raw)
    opt = torch.optim.Adam(model.param_groups(), lr=0.0, eps=1e-15)
    grads = [torch.zeros_like(p) for p in model.parameters()]
    accum = torch.zeros(120, 1); denom = torch.zeros(120, 1); maxr = torch.zeros(120)
    for v in (0, 1):
        pkg = _oracle_render(cams[v], model, PipelineParams(), torch.zeros(3))
        loss = training_loss(pkg["render"], gts[v])
        loss.backward()
        vis = pkg["visibility_filter"]
        accum[vis] += torch.norm(pkg["viewspace_points"].grad[vis, :2], dim=-1, keepdim=True)
        denom[vis] += 1
        maxr[vis] = torch.max(maxr[vis], pkg["radii"][vis].float())
        for g, p in zip(grads, model.parameters()):
            g += p.grad
            p.grad = None
    for g, p in zip(grads, model.parameters()):
        p.grad = g / 2
    opt.step()
    for pa, p in zip(a["params"], model.parameters()):
        assert torch.allclose(pa, p.detach(), atol=1e-6, rtol=1e-5)
    assert torch.allclose(a["accum"], accum, atol=1e-6) and torch.equal(a["denom"], denom) and torch.equal(a["maxr"], maxr)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("exchange", ["sharded", "visible_rows"])
def test_sharded_and_visible_row_exchanges_equal_the_all_reduce_schedule(tmp_path, exchange):
    """reduce-scatter -> Adam on a 1/N row shard -> all-gather, and the visible-rows-only all-reduce, against the plain
    all-reduce schedule over two optimizer steps (so the second step sees the first one's moments): identical parameters on
    both ranks and across schedules, bit for bit; the sharded schedule keeps half of the Adam moments per rank.  120 rows on
    2 ranks and, for the left-over rows' path, 121 rows."""
    outs = {}
    for k, ex in enumerate(("allreduce", exchange)):
        d = tmp_path / ex
        d.mkdir()
        port = 31000 + (os.getpid() % 2000) + 7 * k
        mp.spawn(_worker, args=(2, port, str(d), ex, 2), nprocs=2, join=True)
        outs[ex] = [torch.load(os.path.join(d, f"r{r}.pt")) for r in range(2)]
    for ex, (a, b) in outs.items():
        for pa, pb in zip(a["params"], b["params"]):
            assert torch.equal(pa, pb), ex
    for pa, pb in zip(outs["allreduce"][0]["params"], outs[exchange][0]["params"]):
        assert torch.equal(pa, pb)
    if exchange == "sharded":
        full = sum(p.numel() for p in outs["allreduce"][0]["params"]) * 4 * 2
        assert outs["sharded"][0]["moment_bytes"] == full // 2


@pytest.mark.timeout(900)
def test_rank1_sh_exchange_equals_the_all_reduce_schedule_to_rounding(tmp_path):
    """exchange="sh_rank1": geometry gradients all-reduced, the SH gradients rebuilt on every rank from an all-gather of
    dL/df_dc + camera centres (rank one per Gaussian and view).  Two optimizer steps on two ranks: both ranks bit-identical
    (the rebuild sums the ranks in order), and equal to the all-reduce schedule up to fp32 rounding of the individual products -
    tolerance: |delta| <= 1e-7 + 1e-5 |p| on every parameter after the two Adam steps (measured: ~1e-9)."""
    outs = {}
    for k, ex in enumerate(("allreduce", "sh_rank1")):
        d = tmp_path / ex
        d.mkdir()
        port = 35000 + (os.getpid() % 2000) + 7 * k
        mp.spawn(_worker, args=(2, port, str(d), ex, 2), nprocs=2, join=True)
        outs[ex] = [torch.load(os.path.join(d, f"r{r}.pt")) for r in range(2)]
    for ex, (a, b) in outs.items():
        for pa, pb in zip(a["params"], b["params"]):
            assert torch.equal(pa, pb), ex
    moved = False
    for pa, pb in zip(outs["allreduce"][0]["params"], outs["sh_rank1"][0]["params"]):
        assert torch.allclose(pa, pb, atol=1e-7, rtol=1e-5), float((pa - pb).abs().max())
        moved |= bool((pa != pb).any())
    # the SH tensors did go through the rebuilt gradients (f_rest is parameter 2)
    raw, _, _ = _scene()
    assert not torch.equal(outs["sh_rank1"][0]["params"][2], raw.features_rest)


def test_exchange_bytes_table():
    from scene_utils import exchange_bytes_per_gaussian as b
    assert b("allreduce", 8) == pytest.approx(2 * 7 / 8 * 236)
    assert b("sharded", 8) == pytest.approx(b("allreduce", 8))
    assert b("sh_rank1", 8) == pytest.approx(2 * 7 / 8 * 44 + 7 * 12)
    assert b("sh_rank1", 8) < 0.4 * b("allreduce", 8)
    assert b("visible_rows", 8, visible_fraction=0.25) < 0.3 * b("allreduce", 8)


def test_sharded_step_rccl_branches_with_an_in_process_group(monkeypatch):
    """ShardedStep's RCCL-only branches (reduce_scatter_tensor with AVG, the in-place all_gather_into_tensor whose input is a
    slice of its own output, the AVG all-reduce of the left-over rows) never run under gloo.  Here two threads play the two
    ranks over an in-process stand-in for the collectives that reports itself as "nccl", 121 rows (60-row shards + one
    left-over row), three steps: both ranks must end with the parameters of a single-process Adam on the mean gradients."""
    import threading
    for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-slam_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from scene_utils import GaussianModel, ShardedStep, make_gaussians
    world = 2
    tl = threading.local()
    barrier = threading.Barrier(world)
    slots = [None] * world

    class Work:
        def wait(self):
            return True

    def everyone(t):
        slots[tl.rank] = t.detach().clone()          # (clone: the in-place gather's input aliases its output)
        barrier.wait()
        vals = list(slots)
        barrier.wait()
        return vals

    def reduce_scatter_tensor(out, inp, op=None, group=None, async_op=False):
        assert op == dist.ReduceOp.SUM               # (ShardedStep scales the shard itself: see scene_utils/parallel.py)
        vals = everyone(inp)
        c = out.shape[0]
        out.copy_(sum(v[tl.rank * c:(tl.rank + 1) * c] for v in vals))
        return Work()

    def all_reduce(t, op=None, group=None, async_op=False):
        assert op == dist.ReduceOp.AVG
        vals = everyone(t)
        t.copy_(sum(vals) / world)
        return Work()

    def all_gather_into_tensor(out, inp, group=None, async_op=False):
        vals = everyone(inp)
        out.copy_(torch.cat(vals, dim=0))
        return Work()

    monkeypatch.setattr(dist, "get_backend", lambda group=None: "nccl")
    monkeypatch.setattr(dist, "reduce_scatter_tensor", reduce_scatter_tensor)
    monkeypatch.setattr(dist, "all_reduce", all_reduce)
    monkeypatch.setattr(dist, "all_gather_into_tensor", all_gather_into_tensor)
    P, steps = 121, 3
    results, errors = [None] * world, []

    def grads_of(it, rank, model):
        gen = torch.Generator().manual_seed(1000 * it + rank)
        return [torch.randn(p.shape, generator=gen) for p in model.parameters()]

    def run(rank):
        try:
            tl.rank = rank
            model = GaussianModel.from_raw(make_gaussians(P, 1, seed=5))
            st = ShardedStep(model, lambda groups: torch.optim.Adam(groups, lr=0.0, eps=1e-15), world, rank)
            for it in range(steps):
                for p, g in zip(model.parameters(), grads_of(it, rank, model)):
                    p.grad = g
                st.step()
            results[rank] = [p.detach().clone() for p in model.parameters()]
        except Exception as e:      # pragma: no cover
            errors.append(e)
            barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    model = GaussianModel.from_raw(make_gaussians(P, 1, seed=5))
    opt = torch.optim.Adam(model.param_groups(), lr=0.0, eps=1e-15)
    for it in range(steps):
        gs = [grads_of(it, r, model) for r in range(world)]
        for i, p in enumerate(model.parameters()):
            p.grad = sum(g[i] for g in gs) / world
        opt.step()
    for a, b, ref in zip(results[0], results[1], model.parameters()):
        assert torch.equal(a, b)
        assert torch.equal(a, ref.detach())


def test_sharded_step_handles_rows_not_divisible_by_world(tmp_path):
    """P = 121 on 2 ranks: 60-row shards plus one left-over row that both ranks update from the all-reduced gradient."""
    d = tmp_path / "odd"
    d.mkdir()
    port = 33000 + (os.getpid() % 2000)
    mp.spawn(_worker_odd, args=(2, port, str(d)), nprocs=2, join=True)
    a, b = (torch.load(os.path.join(d, f"r{r}.pt")) for r in range(2))
    for x, y, z in zip(a["sharded"], b["sharded"], a["allreduce"]):
        assert torch.equal(x, y) and torch.equal(x, z)


def _worker_odd(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-slam_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from scene_utils import init_from_env, GaussianModel, GradBucket, ShardedStep, make_gaussians
    init_from_env("gloo")
    res = {}
    for mode in ("allreduce", "sharded"):
        model = GaussianModel.from_raw(make_gaussians(121, 1, seed=5))
        mk = lambda groups: torch.optim.Adam(groups, lr=0.0, eps=1e-15)      # noqa: E731
        if mode == "sharded":
            st = ShardedStep(model, mk, world, rank)
        else:
            opt, bucket = mk(model.param_groups()), GradBucket(model.parameters())
        for it in range(3):
            gen = torch.Generator().manual_seed(1000 * it + rank)
            for p in model.parameters():
                p.grad = torch.randn(p.shape, generator=gen)
            if mode == "sharded":
                st.step()
            else:
                bucket.all_reduce_mean(world)
                opt.step()
                opt.zero_grad(set_to_none=True)
        res[mode] = [p.detach().clone() for p in model.parameters()]
    torch.save(res, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_adjacent_spans_grouping_is_address_independent():
    """Gradients returned side by side in one allocation travel as one collective; the grouping (and the order in which spans
    come out = the order collectives are issued in) depends on list order and offsets only, never on addresses."""
    sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-slam_amd"))
    from scene_utils.parallel import adjacent_spans
    from diff_gaussian_rasterization import _grad_arena, GRAD_ARENA_ALIGN
    P = 1001
    xyz, op, sc, rot, dc, rest, col, cov = _grad_arena("cpu", (((P, 3), 0), ((P, 1), 0), ((P, 3), 0), ((P, 4), 0), ((P, 1, 3), 3),
                                                               ((P, 15, 3), 0), (None, 0), (None, 0)))
    assert col is None and cov is None
    for t in (xyz, op, sc, rot, dc, rest):
        assert t.is_contiguous() and t.storage_offset() % GRAD_ARENA_ALIGN == 0
    assert dc.storage_offset() + dc.numel() + 3 <= rest.storage_offset()        # the spare row behind dc
    lone = torch.zeros(7)
    spans = adjacent_spans([lone, xyz, op, sc, rot])
    assert [len(m) for _, m in spans] == [1, 4] and spans[1][1][0] is xyz and spans[1][0].numel() == rot.storage_offset() + rot.numel()
    # param order of the model (xyz, f_dc, f_rest, opacity, scaling, rotation): still one span, members by offset
    flat, members = adjacent_spans([xyz, dc, rest, op, sc, rot])[0]
    assert [m.data_ptr() for m in members] == [t.data_ptr() for t in (xyz, op, sc, rot, dc, rest)]
    for i, t in enumerate((xyz, op, sc, rot, dc, rest)):
        t.fill_(float(i + 1))
    flat.mul_(2.0)
    assert all(bool((t == 2.0 * (i + 1)).all()) for i, t in enumerate((xyz, op, sc, rot, dc, rest)))
    # a hole larger than the tolerated gap splits the span: geometry without opacity ... rotation
    assert [len(m) for _, m in adjacent_spans([xyz, rot])] == [1, 1]
    # different storages keep list order
    a, b = torch.zeros(4), torch.zeros(3)
    assert [f.numel() for f, _ in adjacent_spans([a, b])] == [4, 3] and [f.numel() for f, _ in adjacent_spans([b, a])] == [3, 4]


def test_all_reduce_mean_of_arena_gradients_two_ranks(tmp_path):
    """GradBucket.all_reduce_mean with the gradients in one arena (one collective over the span) gives the per-tensor mean."""
    port = 35000 + (os.getpid() % 2000)
    mp.spawn(_worker_arena, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = (torch.load(os.path.join(tmp_path, f"r{r}.pt")) for r in range(2))
    for x, y, m in zip(a["got"], b["got"], a["expect"]):
        assert torch.equal(x, y) and torch.equal(x, m)
    assert a["spans"] == b["spans"] == 1


def _worker_arena(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-slam_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from scene_utils import init_from_env, GradBucket
    from scene_utils.parallel import adjacent_spans
    from diff_gaussian_rasterization import _grad_arena
    init_from_env("gloo")
    P = 333
    shapes = ((P, 3), (P, 1), (P, 3), (P, 4))
    params = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
    per_rank = []
    for r in range(world):
        gen = torch.Generator().manual_seed(50 + r)
        per_rank.append([torch.randn(s, generator=gen) for s in shapes])
    grads = _grad_arena("cpu", tuple((s, 0) for s in shapes))
    for p, g, src in zip(params, grads, per_rank[rank]):
        g.copy_(src)
        p.grad = g
    n_spans = len(adjacent_spans([p.grad for p in params]))
    GradBucket(params).all_reduce_mean(world)
    expect = [(a + b) * (1.0 / world) for a, b in zip(*per_rank)]
    torch.save(dict(got=[p.grad.clone() for p in params], expect=expect, spans=n_spans), os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()
