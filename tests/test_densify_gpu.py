"""GPU parity of densify_and_prune: HIP plan/apply kernels against the CPU restatement of reference
scene/gaussian_model.py:367-429 (oracle/densify_oracle.py).  Deterministic structure (which rows survive, are cloned, are split,
their order, copied values, Adam moments) must match exactly; the split's normal samples are checked statistically."""
import math

import pytest
import torch

from oracle import densify_oracle as DO
from scene_utils import make_gaussians, GaussianModel

pytestmark = pytest.mark.gpu

NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
ATTRS = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")


def _setup(P=6000, seed=3, with_moments=True, deg=3):
    raw = make_gaussians(P, deg, seed=seed, scale_factor=1.0)
    gen = torch.Generator().manual_seed(seed + 1)
    raw.scaling += 1.2 * torch.randn(P, 3, generator=gen)                 # wide spread around percent_dense * extent
    raw.opacity[torch.rand(P, generator=gen) < 0.15] = -7.0              # some below min_opacity
    model = GaussianModel.from_raw(raw.to("cuda"))
    opt = model.training_setup(optimizer="hip")
    if with_moments:
        for p in model.parameters():
            p.grad = torch.randn(p.shape, generator=gen).cuda() * 1e-3
        opt.step()
        opt.zero_grad(set_to_none=True)
    accum = torch.rand(P, 1, generator=gen) * 0.002
    den = torch.randint(0, 4, (P, 1), generator=gen).float()
    accum[den == 0] = 0.0                                                 # 0/0 -> NaN -> 0 path
    model.xyz_gradient_accum, model.denom = accum.cuda(), den.cuda()
    model.max_radii2D = (torch.rand(P, generator=gen) * 50).cuda()
    return model, opt, accum, den


@pytest.mark.parametrize("max_screen_size", [None, 20])
def test_densify_matches_reference_semantics(max_screen_size):
    model, opt, accum, den = _setup()
    P = accum.shape[0]
    params = {n: getattr(model, a).detach().cpu().clone() for n, a in zip(NAMES, ATTRS)}
    moments = {}
    for n, a in zip(NAMES, ATTRS):
        st = opt.state[getattr(model, a)]
        moments[n] = (st["exp_avg"].cpu().clone(), st["exp_avg_sq"].cpu().clone())
    extent, thr, min_op = 15.0, 0.0004, 0.005     # percent_dense*extent = 0.15 splits the scale distribution
    ref_p, ref_m, info = DO.densify_and_prune(params, moments, accum.clone(), den.clone(), model.max_radii2D.cpu().clone(), thr,
                                              min_op, extent, max_screen_size, model.percent_dense,
                                              normal_samples=torch.zeros(0, 3) if False else None)
    nk, nc, ns, src = model.densify_and_prune(thr, min_op, extent, max_screen_size, None, seed=11, return_source=True)
    src = src.cpu().long()
    kind = info["kind"]
    assert (nk, nc, 2 * ns) == (int((kind == 0).sum()), int((kind == 1).sum()), int((kind == 2).sum()))
    assert nk > 100 and nc > 100 and ns > 100                                  # the scene exercises every branch
    assert torch.equal(src, info["source"])                                    # same rows, same order
    newP = nk + nc + 2 * ns
    assert model.get_xyz.shape[0] == newP and model.xyz_gradient_accum.shape == (newP, 1)
    assert float(model.xyz_gradient_accum.abs().sum()) == 0 and float(model.max_radii2D.abs().sum()) == 0
    det = kind != 2                                                            # kept + clones: everything exact
    for n, a in zip(NAMES, ATTRS):
        got = getattr(model, a).detach().cpu()
        assert got.shape == ref_p[n].shape
        assert torch.equal(got[det], ref_p[n][det]), n
        st = opt.state[getattr(model, a)]
        assert torch.equal(st["exp_avg"].cpu(), ref_m[n][0]) and torch.equal(st["exp_avg_sq"].cpu(), ref_m[n][1]), n
        for g in opt.param_groups:
            if g["name"] == n:
                assert g["params"][0] is getattr(model, a)
    ch = kind == 2
    for n in ("f_dc", "f_rest", "opacity", "rotation"):
        assert torch.equal(getattr(model, ATTRS[NAMES.index(n)]).detach().cpu()[ch], ref_p[n][ch]), n
    assert torch.allclose(model._scaling.detach().cpu()[ch], ref_p["scaling"][ch], atol=2e-6)
    # split samples: z = R^T (xyz_child - xyz_src) / exp(scaling_src)  ~  N(0,1)
    s_idx = src[ch]
    R = DO.build_rotation(params["rotation"][s_idx])
    d = model._xyz.detach().cpu()[ch] - params["xyz"][s_idx]
    z = torch.bmm(R.transpose(1, 2), d.unsqueeze(-1)).squeeze(-1) / torch.exp(params["scaling"][s_idx])
    n = z.numel()
    assert abs(float(z.mean())) < 5.0 / math.sqrt(n) and abs(float(z.std()) - 1.0) < 0.05
    assert float((z.abs() > 4.5).float().mean()) < 1e-3
    half = z.shape[0] // 2
    assert not torch.allclose(z[:half], z[half:])                              # the two copies are independent draws
    assert abs(float((z[:half] * z[half:]).mean())) < 0.05


def test_densify_without_optimizer_state_and_reset_opacity():
    model, opt, accum, den = _setup(P=1500, with_moments=False)
    before = model.get_opacity.detach().clone()
    nk, nc, ns = model.densify_and_prune(0.0004, 0.005, 2.0, None)
    assert model.get_xyz.shape[0] == nk + nc + 2 * ns
    model.reset_opacity()
    assert float(model.get_opacity.max()) <= 0.01 + 1e-6
    for g in opt.param_groups:
        if g["name"] == "opacity":
            assert g["params"][0] is model._opacity
    # training continues on the new parameter set
    for p in model.parameters():
        p.grad = torch.ones_like(p) * 1e-3
    opt.step()
    assert torch.isfinite(model._xyz).all()


def test_training_with_densification_schedule():
    """Trainer end-to-end with the reference schedule compressed (densify every 5 its from it 5): the row count changes,
    rendering / loss / Adam keep working, and the loss stays finite."""
    from gaussian_renderer import render, PipelineParams
    from scene_utils import fibonacci_cameras, Trainer
    raw = make_gaussians(4000, 1, seed=8, scale_factor=0.8)
    cams = fibonacci_cameras(3, 160, 96, seed=9, device="cuda")
    teacher = GaussianModel.from_raw(make_gaussians(4000, 1, seed=10, scale_factor=0.8).to("cuda"), requires_grad=False)
    bg = torch.zeros(3, device="cuda")
    pipe = PipelineParams()
    with torch.no_grad():
        gts = {i: render(c, teacher, pipe, bg)["render"].clone() for i, c in enumerate(cams)}
    model = GaussianModel.from_raw(raw.to("cuda"))
    tr = Trainer(model, cams, gts, render, pipe, bg, separate_sh=True)
    tr.enable_densification(extent=2.0, from_iter=4, until_iter=40, interval=5, opacity_reset_interval=15, grad_threshold=1e-5)
    sizes = []
    for it in range(30):
        out = tr.step(it % 3)
        sizes.append(model.get_xyz.shape[0])
        assert torch.isfinite(out["loss"])
    tr.finish()
    assert len(set(sizes)) > 2
    assert model.xyz_gradient_accum.shape[0] == sizes[-1]


def test_densification_stats_kernel_matches_reference_lines():
    """k_densify_stats against reference scene/gaussian_model.py:431-433 + train.py:159 written with torch ops."""
    gen = torch.Generator().manual_seed(5)
    P = 5003
    m = GaussianModel.from_raw(make_gaussians(P, 0, seed=1).to("cuda"))
    m.training_setup(optimizer="hip")
    m.xyz_gradient_accum = torch.rand(P, 1, generator=gen).cuda()
    m.denom = torch.randint(0, 5, (P, 1), generator=gen).float().cuda()
    m.max_radii2D = (torch.rand(P, generator=gen) * 30).cuda()
    acc0, den0, mr0 = m.xyz_gradient_accum.clone(), m.denom.clone(), m.max_radii2D.clone()
    radii = torch.randint(0, 60, (P,), generator=gen, dtype=torch.int32)
    radii[torch.rand(P, generator=gen) < 0.4] = 0
    radii = radii.cuda()
    vsp = torch.zeros(P, 3, device="cuda", requires_grad=True)
    vsp.grad = torch.randn(P, 3, generator=gen).cuda()
    vis = radii > 0
    m.add_densification_stats(vsp, vis, radii)
    acc0[vis] += torch.norm(vsp.grad[vis, :2], dim=-1, keepdim=True)            # gaussian_model.py:432
    den0[vis] += 1                                                              # :433
    mr0[vis] = torch.max(mr0[vis], radii[vis].float())                          # train.py:159
    assert torch.allclose(m.xyz_gradient_accum, acc0, rtol=1e-6, atol=1e-7)
    assert torch.equal(m.denom, den0) and torch.equal(m.max_radii2D, mr0)


def test_densification_stats_folded_into_backward_match_the_separate_kernel():
    """gsr_grads.xyz_gradient_accum / denom / max_radii2D: the rasterizer's backward updating the statistics itself against
    the separate pass (k_densify_stats on viewspace_points.grad), bit for bit, over three views (accumulation included)."""
    import diff_gaussian_rasterization as dgr
    from gaussian_renderer import render, PipelineParams
    from scene_utils import fibonacci_cameras
    cams = fibonacci_cameras(3, 160, 96, seed=77, device="cuda")
    bg = torch.zeros(3, device="cuda")
    res = {}
    for folded in (False, True):
        m = GaussianModel.from_raw(make_gaussians(4000, 2, seed=78, scale_factor=0.7).to("cuda"))
        m.training_setup(optimizer="hip")
        for i, cam in enumerate(cams):
            fold = dgr.BackwardFold(stats=(m.xyz_gradient_accum, m.denom, m.max_radii2D)) if folded else None
            pkg = render(cam, m, PipelineParams(), bg, separate_sh=True, fold=fold)
            (pkg["render"] * (1.0 + i)).sum().backward()
            if folded:
                assert fold.stats_taken and not fold.optimizer_taken
            else:
                m.add_densification_stats(pkg["viewspace_points"], pkg["visibility_filter"], pkg["radii"])
            for p in m.parameters():
                p.grad = None
        torch.cuda.synchronize()
        res[folded] = (m.xyz_gradient_accum.clone(), m.denom.clone(), m.max_radii2D.clone())
    assert float(res[True][1].sum()) > 1000
    for a, b in zip(res[False], res[True]):
        assert torch.equal(a, b)


def test_densify_edge_cases_nothing_survives_and_sh_degree_zero():
    """(a) every row below min_opacity and no gradient: the model ends up EMPTY (reference: prune_points with an all-true mask)
    and still renders (background); (b) an SH-degree-0 model - f_rest is [P, 0, 3], no storage - densifies like any other.
    Both were found by tests/sweeps/extended_densify_sweep.py (the apply call refused the NULL pointers of empty tensors)."""
    from gaussian_renderer import render, PipelineParams
    from scene_utils import fibonacci_cameras
    model, opt, accum, den = _setup(P=37, seed=9)
    with torch.no_grad():
        model._opacity.fill_(-9.0)
    model.xyz_gradient_accum.zero_()
    nk, nc, ns = model.densify_and_prune(0.0004, 0.005, 15.0, None, None, seed=1)
    assert (nk, nc, ns) == (0, 0, 0) and model.get_xyz.shape[0] == 0 and model._features_rest.shape == (0, 15, 3)
    for g in opt.param_groups:
        assert g["params"][0].shape[0] == 0
    cam = fibonacci_cameras(1, 64, 48, seed=2, device="cuda")[0]
    bg = torch.tensor([0.3, 0.2, 0.1], device="cuda")
    img = render(cam, model, PipelineParams(), bg)["render"]
    assert torch.equal(img, bg[:, None, None].expand_as(img))
    # (b)
    model, opt, accum, den = _setup(P=4001, seed=10, deg=0)
    assert model._features_rest.numel() == 0
    params = {n: getattr(model, a).detach().cpu().clone() for n, a in zip(NAMES, ATTRS)}
    moments = {}
    for n, a in zip(NAMES, ATTRS):
        st = opt.state.get(getattr(model, a), {})
        moments[n] = (st["exp_avg"].cpu().clone(), st["exp_avg_sq"].cpu().clone()) if "exp_avg" in st else None
    ref_p, ref_m, info = DO.densify_and_prune(params, moments, accum.clone(), den.clone(), model.max_radii2D.cpu().clone(), 0.0004,
                                              0.005, 15.0, 20, model.percent_dense, normal_samples=None)
    nk, nc, ns, src = model.densify_and_prune(0.0004, 0.005, 15.0, 20, None, seed=3, return_source=True)
    kind = info["kind"]
    assert (nk, nc, 2 * ns) == (int((kind == 0).sum()), int((kind == 1).sum()), int((kind == 2).sum())) and nc > 10 and ns > 10
    assert torch.equal(src.cpu().long(), info["source"])
    det = kind != 2
    for n, a in zip(NAMES, ATTRS):
        got = getattr(model, a).detach().cpu()
        assert got.shape == ref_p[n].shape and torch.equal(got[det], ref_p[n][det]), n


@pytest.mark.parametrize("kind", ["hip", "hip_sparse", "hip_fused", "hip_sparse_fused"])
def test_training_goes_on_after_the_model_was_pruned_to_empty(kind):
    """min_opacity above every opacity: the first densification prunes the model to ZERO rows; the steps after it (render = the
    background, empty gradients, an optimizer step over nothing) must simply run (found by tests/sweeps/extended_fused_sweep.py
    --densify: SparseGaussianAdam's launch refused N = 0)."""
    from gaussian_renderer import render, PipelineParams
    from scene_utils import fibonacci_cameras, Trainer
    cams = fibonacci_cameras(2, 96, 64, seed=4, device="cuda")
    bg = torch.tensor([0.1, 0.2, 0.3], device="cuda")
    gts = {i: torch.rand(3, 64, 96, device="cuda") for i in range(2)}
    model = GaussianModel.from_raw(make_gaussians(500, 2, seed=6, scale_factor=0.8).to("cuda"))
    tr = Trainer(model, cams, gts, render, PipelineParams(), bg, separate_sh=True, optimizer=kind)
    tr.enable_densification(extent=4.4, from_iter=1, until_iter=100, interval=2, opacity_reset_interval=50, grad_threshold=1e-5,
                            min_opacity=1.1)
    for it in range(6):
        tr.step(it % 2)
    tr.finish()
    torch.cuda.synchronize()
    assert model.get_xyz.shape[0] == 0
    assert torch.equal(tr.last["image"], bg[:, None, None].expand(3, 64, 96))
