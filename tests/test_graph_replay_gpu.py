"""The single-view training step as ONE HIP graph launch (Trainer.enable_graph_replay): forward, loss, backward with the folded
optimizer and the densification statistics are captured once and replayed; camera, ground truth and the optimizer's per-step
factors live in device memory.  Everything must equal the eager loop bit for bit - also when a replayed frame exceeds the
capacity the graph was captured for (device-side no-op, eager re-run, new capture)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(kind, graph, steps=14, shrink_capacity_at=None, depth=False):
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _workspace as ws
    from gaussian_renderer import render, PipelineParams
    from scene_utils import make_gaussians, fibonacci_cameras, GaussianModel, Trainer
    dev = "cuda"
    cams = fibonacci_cameras(3, 208, 128, seed=41, device=dev)
    bg = torch.tensor([0.05, 0.1, 0.2], device=dev)
    pipe = PipelineParams()
    teacher = GaussianModel.from_raw(make_gaussians(6000, 2, seed=42, scale_factor=0.7).to(dev), requires_grad=False)
    with torch.no_grad():
        pk = [render(c, teacher, pipe, bg) for c in cams]
        gts = {i: p["render"].clone() for i, p in enumerate(pk)}
        dts = {i: p["depth"].clone() for i, p in enumerate(pk)} if depth else None
    model = GaussianModel.from_raw(make_gaussians(6000, 2, seed=43, scale_factor=0.7).to(dev))
    ws.pool(torch.device(dev, 0)).forget_estimates()
    tr = Trainer(model, cams, gts, render, pipe, bg, separate_sh=True, optimizer=kind, depth_targets=dts,
                 depth_weight=0.5 if depth else 0.0)
    if graph:
        tr.enable_graph_replay()
    losses = []
    for it in range(steps):
        if graph and shrink_capacity_at == it:
            pool = ws.pool(torch.device(dev, 0))
            for k in list(pool.capacity):
                pool.capacity[k] = ws.MIN_CAPACITY        # far below what the frames need: the graph captured next overflows
            tr._graph, tr._graph_sig = None, None
        tr.step(it % 3)
        losses.append(tr.last["loss"].clone())
    tr.finish()
    torch.cuda.synchronize()
    out = []
    for p in model.parameters():
        st = tr.optimizer.state[p]
        out += [p.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone()]
    out += [model.xyz_gradient_accum.clone(), model.denom.clone(), model.max_radii2D.clone()]
    return out, torch.stack(losses).cpu(), getattr(tr, "graph_stats", None), tr.iteration


@pytest.mark.parametrize("kind", ["hip_fused", "hip_sparse_fused"])
def test_graph_replay_equals_the_eager_loop(kind):
    ref, l_ref, _, it_ref = _run(kind, False)
    got, l_got, stats, it_got = _run(kind, True)
    assert stats["captures"] == 1 and stats["replays"] == 12 and stats["eager_steps"] == 2 and stats["overflow_reruns"] == 0
    assert it_ref == it_got == 14
    for i, (a, b) in enumerate(zip(ref, got)):
        assert a.shape == b.shape and torch.equal(a, b), (i, float((a - b).abs().max()))
    assert torch.equal(l_ref, l_got)


def test_graph_replay_with_the_depth_term():
    ref, l_ref, _, _ = _run("hip_fused", False, depth=True)
    got, l_got, stats, _ = _run("hip_fused", True, depth=True)
    assert stats["replays"] == 12
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    assert torch.equal(l_ref, l_got)


def test_replayed_frame_beyond_the_graphs_capacity_is_rerun_and_the_graph_recaptured():
    ref, _, _, it_ref = _run("hip_fused", False)
    got, _, stats, it_got = _run("hip_fused", True, shrink_capacity_at=6)
    assert stats["overflow_reruns"] == 1 and stats["captures"] == 3 and it_ref == it_got
    for i, (a, b) in enumerate(zip(ref, got)):
        assert torch.equal(a, b), i


@pytest.mark.parametrize("min_opacity", [0.005, 1.1])
def test_graph_replay_across_densifications(min_opacity):
    """A densification changes the model size: the steps around it run eagerly, the graph is captured anew for the new size -
    and a model pruned to EMPTY (min_opacity above every opacity) simply goes on eagerly.  Same end state as the eager loop."""
    from gaussian_renderer import render, PipelineParams
    from scene_utils import make_gaussians, fibonacci_cameras, GaussianModel, Trainer
    dev = "cuda"
    cams = fibonacci_cameras(3, 160, 96, seed=51, device=dev)
    bg = torch.zeros(3, device=dev)
    pipe = PipelineParams()
    teacher = GaussianModel.from_raw(make_gaussians(3000, 1, seed=52, scale_factor=0.8).to(dev), requires_grad=False)
    with torch.no_grad():
        gts = {i: render(c, teacher, pipe, bg)["render"].clone() for i, c in enumerate(cams)}
    ends = []
    for graph in (False, True):
        model = GaussianModel.from_raw(make_gaussians(3000, 1, seed=53, scale_factor=0.8).to(dev))
        tr = Trainer(model, cams, gts, render, pipe, bg, separate_sh=True, optimizer="hip_fused")
        tr.enable_densification(extent=4.4, from_iter=3, until_iter=100, interval=6, opacity_reset_interval=50,
                                grad_threshold=2e-5, min_opacity=min_opacity, seed=7)
        if graph:
            tr.enable_graph_replay(warmup=1)
        for it in range(20):
            tr.step(it % 3)
        tr.finish()
        torch.cuda.synchronize()
        ends.append(([p.detach().clone() for p in model.parameters()], tr.graph_stats if graph else None))
    (ref, _), (got, stats) = ends
    for a, b in zip(ref, got):
        assert a.shape == b.shape and torch.equal(a, b)
    if min_opacity < 1:
        assert stats["captures"] >= 3 and stats["replays"] >= 8 and ref[0].shape[0] > 3000
    else:
        assert ref[0].shape[0] == 0 and stats["captures"] == 1


def test_opacity_reset_between_two_replays_drops_the_graph():
    """`reset_opacity` replaces `_opacity` and its two Adam moments and leaves `_xyz` / `_features_rest` alone.  With
    the model at `max_gaussians` the reset iterations (5, 10, 15) change nothing else, so a signature of sizes + those two
    addresses kept replaying a graph whose baked pointers were the OLD opacity tensor and moments (freed memory; the real
    `_opacity` never trained again).  The graph is now dropped whenever densification / reset changed anything, and the signature
    covers every parameter, moment and statistics address: same end state as the eager loop, bit for bit, moments included."""
    from gaussian_renderer import render, PipelineParams
    from scene_utils import make_gaussians, fibonacci_cameras, GaussianModel, Trainer
    dev = "cuda"
    cams = fibonacci_cameras(3, 160, 96, seed=61, device=dev)
    bg = torch.zeros(3, device=dev)
    pipe = PipelineParams()
    teacher = GaussianModel.from_raw(make_gaussians(3000, 1, seed=62, scale_factor=0.8).to(dev), requires_grad=False)
    with torch.no_grad():
        gts = {i: render(c, teacher, pipe, bg)["render"].clone() for i, c in enumerate(cams)}
    ends = []
    for graph in (False, True):
        model = GaussianModel.from_raw(make_gaussians(3000, 1, seed=63, scale_factor=0.8).to(dev))
        tr = Trainer(model, cams, gts, render, pipe, bg, separate_sh=True, optimizer="hip_fused", lambda_dssim=0.2)
        tr.enable_densification(extent=4.4, from_iter=3, until_iter=100, interval=6, opacity_reset_interval=5,
                                grad_threshold=2e-5, min_opacity=0.005, seed=7, max_gaussians=3000)
        if graph:
            tr.enable_graph_replay(warmup=1)
        opacity_ptrs = set()
        for it in range(19):
            tr.step(it % 3)
            opacity_ptrs.add(model._opacity.data_ptr())
        tr.finish()
        torch.cuda.synchronize()
        out = []
        for p in model.parameters():
            st = tr.optimizer.state[p]
            out += [p.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone()]
        ends.append((out, tr.graph_stats if graph else None, len(opacity_ptrs), int(model.get_xyz.shape[0])))
    (ref, _, n_ref, P_ref), (got, stats, n_got, P_got) = ends
    assert P_ref == P_got == 3000 and n_ref >= 2 and n_got >= 2          # the resets did replace the opacity tensor
    assert stats["captures"] >= 4 and stats["replays"] >= 8               # one capture per stretch between resets
    for i, (a, b) in enumerate(zip(ref, got)):
        assert torch.equal(a, b), (i, float((a - b).abs().max()))
    # the scalars baked into a captured step are part of the signature too
    tr.lambda_dssim = 0.3
    sig_a = tr._graph_signature(cams[0])
    tr.lambda_dssim = 0.2
    assert sig_a != tr._graph_signature(cams[0])
