"""GPU parity of the two callers of the hot path that the reference also takes from native modules:
fused SSIM (oracle: the pure-PyTorch ssim() restated in oracle/loss_oracle.py, itself pinned against the reference's
utils/loss_utils.py:ssim by tests/golden/reference_helpers.npz) and the one-launch Adam kernels (oracle: torch.optim.Adam
on CPU; the sparse variant against a masked no-bias-correction restatement)."""
import numpy as np
import pytest
import torch

from oracle import loss_oracle as losses

pytestmark = pytest.mark.gpu


# (3, 601, 1203): 4332 tiles, past the 4096 at which a forward workgroup walks two tiles with the next halo in flight
@pytest.mark.parametrize("shape", [(3, 40, 52), (1, 3, 67, 33), (3, 1080 // 4, 1920 // 4), (2, 3, 16, 16), (3, 601, 1203)])
def test_fused_ssim_matches_reference_ssim(shape):
    from fused_ssim import fused_ssim
    gen = torch.Generator().manual_seed(sum(shape))
    a = torch.rand(*shape, generator=gen, dtype=torch.float64)
    b = (a + 0.1 * torch.randn(*shape, generator=gen, dtype=torch.float64)).clamp(0, 1)
    a_ref = a.clone().requires_grad_(True)
    val_ref = losses.ssim(a_ref, b)
    val_ref.backward()
    a_gpu = a.float().cuda().requires_grad_(True)
    val = fused_ssim(a_gpu if a.dim() == 4 else a_gpu.unsqueeze(0), b.float().cuda() if a.dim() == 4 else b.float().cuda().unsqueeze(0))
    val.backward()
    assert abs(val.item() - val_ref.item()) <= 2e-6
    g, g_ref = a_gpu.grad.cpu().double(), a_ref.grad
    assert (g - g_ref).norm() / g_ref.norm() <= 1e-4
    assert (g - g_ref).abs().max() <= 1e-4 * g_ref.abs().max() + 1e-9


def test_fused_ssim_golden_and_interfaces():
    """Value against the reference's own ssim() output (golden), and the `_C.fusedssim` / `fusedssim_backward` interface
    of reference utils/loss_utils.py:16-38."""
    import os
    from diff_gaussian_rasterization._C import fusedssim, fusedssim_backward
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_helpers.npz"))
    a, b = torch.tensor(G["img_a2"]).cuda(), torch.tensor(G["img_b2"]).cuda()
    m = fusedssim(0.01 ** 2, 0.03 ** 2, a, b)
    assert m.shape == a.shape
    assert abs(m.mean().item() - float(G["ssim_ab2"])) <= 2e-6
    g = fusedssim_backward(0.01 ** 2, 0.03 ** 2, a, b, torch.full_like(a, 1.0 / a.numel()))
    a_ref = torch.tensor(G["img_a2"], dtype=torch.float64, requires_grad=True)
    losses.ssim(a_ref, torch.tensor(G["img_b2"], dtype=torch.float64)).backward()
    assert (g.cpu().double() - a_ref.grad).norm() / a_ref.grad.norm() <= 1e-4


def test_fused_adam_matches_torch_adam():
    from diff_gaussian_rasterization import FusedAdam
    gen = torch.Generator().manual_seed(0)
    shapes = [(1001, 3), (1001, 1, 3), (1001, 15, 3), (1001, 1), (1001, 3), (1001, 4)]
    lrs = [0.00016, 0.0025, 0.000125, 0.025, 0.005, 0.001]
    ref = [torch.randn(*s, generator=gen).requires_grad_(True) for s in shapes]
    dev = [r.detach().clone().cuda().requires_grad_(True) for r in ref]
    o_ref = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(ref, lrs)], lr=0.0, eps=1e-15)
    o_dev = FusedAdam([{"params": [p], "lr": lr} for p, lr in zip(dev, lrs)], lr=0.0, eps=1e-15)
    for it in range(5):
        for r, d in zip(ref, dev):
            g = torch.randn(*r.shape, generator=gen) * (10.0 ** torch.randint(-6, 1, (1,), generator=gen).item())
            r.grad = g.clone()
            d.grad = g.clone().cuda()
        o_ref.step()
        o_dev.step()
    for r, d in zip(ref, dev):
        err = (d.detach().cpu() - r.detach()).abs().max().item()
        assert torch.allclose(d.detach().cpu(), r.detach(), rtol=2e-6, atol=1e-7), err
    st = o_dev.state[dev[2]]
    assert torch.allclose(st["exp_avg_sq"].cpu(), o_ref.state[ref[2]]["exp_avg_sq"], rtol=1e-5, atol=1e-12)


def test_sparse_adam_touches_only_visible_rows():
    from diff_gaussian_rasterization import SparseGaussianAdam
    gen = torch.Generator().manual_seed(1)
    N = 777
    shapes = [(N, 3), (N, 1, 3), (N, 15, 3), (N, 1), (N, 3), (N, 4)]
    ps = [torch.randn(*s, generator=gen) for s in shapes]
    dev = [p.clone().cuda().requires_grad_(True) for p in ps]
    opt = SparseGaussianAdam([{"params": [p], "lr": 0.01 * (i + 1)} for i, p in enumerate(dev)], lr=0.0, eps=1e-15)
    vis = torch.rand(N, generator=gen) < 0.6
    m = [torch.zeros_like(p) for p in ps]
    v = [torch.zeros_like(p) for p in ps]
    cur = [p.clone() for p in ps]
    for it in range(3):
        gs = [torch.randn(*p.shape, generator=gen) for p in ps]
        for d, g in zip(dev, gs):
            d.grad = g.cuda()
        opt.step(vis.cuda(), N)
        for i, g in enumerate(gs):
            mask = vis.view(N, *([1] * (g.dim() - 1))).expand_as(g)
            m_new = 0.9 * m[i] + 0.1 * g
            v_new = 0.999 * v[i] + 0.001 * g * g
            upd = cur[i] - 0.01 * (i + 1) * m_new / (v_new.sqrt() + 1e-15)
            m[i] = torch.where(mask, m_new, m[i]); v[i] = torch.where(mask, v_new, v[i]); cur[i] = torch.where(mask, upd, cur[i])
    for d, c, p0 in zip(dev, cur, ps):
        err = (d.detach().cpu() - c).abs().max().item()
        assert torch.allclose(d.detach().cpu(), c, rtol=2e-6, atol=1e-7), err
        inv = ~vis
        assert torch.equal(d.detach().cpu()[inv], p0[inv])       # invisible rows bit-identical


@pytest.mark.parametrize("shape,lam", [((3, 67, 45), 0.2), ((3, 270, 480), 0.2), ((3, 33, 32), 0.7), ((3, 601, 1203), 0.2)])
def test_fused_l1_ssim_loss_matches_reference_loss(shape, lam):
    """The fused training loss against reference train.py:114-121 evaluated with the pure-PyTorch l1_loss / ssim restatements
    (float64 on the CPU), value and gradient."""
    from fused_ssim import fused_l1_ssim_loss
    gen = torch.Generator().manual_seed(sum(shape))
    a = torch.rand(*shape, generator=gen, dtype=torch.float64)
    b = (a + 0.1 * torch.randn(*shape, generator=gen, dtype=torch.float64)).clamp(0, 1)
    b[:, :4, :4] = a[:, :4, :4]                              # exact ties: sign(0) = 0 like torch
    a_ref = a.clone().requires_grad_(True)
    ref = losses.training_loss(a_ref, b, lam)
    ref.backward()
    a_gpu = a.float().cuda().requires_grad_(True)
    val = fused_l1_ssim_loss(a_gpu, b.float().cuda(), lam)
    (3.0 * val).backward()                                  # non-unit upstream gradient
    assert abs(val.item() - ref.item()) <= 2e-6
    g, g_ref = a_gpu.grad.cpu().double(), 3.0 * a_ref.grad
    assert (g - g_ref).norm() / g_ref.norm() <= 1e-4
    # bitwise reproducible
    a2 = a.float().cuda().requires_grad_(True)
    v2 = fused_l1_ssim_loss(a2, b.float().cuda(), lam)
    (3.0 * v2).backward()
    assert v2.item() == val.item() and torch.equal(a2.grad, a_gpu.grad)


def test_fold_request_travels_with_its_call_unrelated_renders_in_between_take_nothing():
    """SURVEY 8(b) "no global state, re-entrant": the optimizer / statistics fold is an argument of ONE rasterizer call and lives on
    that call's autograd ctx.  Between arming it (the render) and its backward, a viewer-style render of ANOTHER model with its own
    grad-enabled backward - and a forward-only render of the same model - run untouched: they neither consume the fold nor get
    stepped by it, and the folded step still happens on the right call, bit-identical to backward + optimizer.step()."""
    import diff_gaussian_rasterization as dgr
    from gaussian_renderer import render, PipelineParams
    from scene_utils import make_gaussians, fibonacci_cameras, GaussianModel
    from scene_utils.losses import training_loss_fused
    cams = fibonacci_cameras(3, 160, 96, seed=191, device="cuda")
    bg = torch.tensor([0.1, 0.2, 0.05], device="cuda")
    pipe = PipelineParams()
    gen = torch.Generator().manual_seed(192)
    gt = torch.rand(3, 96, 160, generator=gen).cuda()
    up = torch.randn(3, 96, 160, generator=gen).cuda()

    def viewer_grads():
        other = GaussianModel.from_raw(make_gaussians(1500, 2, seed=194, scale_factor=0.8).to("cuda"))
        pkg = render(cams[1], other, pipe, bg, separate_sh=True)
        (pkg["render"] * up).sum().backward()
        return other, [p.grad.clone() for p in other.parameters()]

    _, plain_viewer = viewer_grads()
    res = {}
    for folded in (False, True):
        model = GaussianModel.from_raw(make_gaussians(2500, 3, seed=193, scale_factor=0.7).to("cuda"))
        opt = model.training_setup(optimizer="hip")
        for it in range(3):
            fold = dgr.BackwardFold(optimizer=opt, stats=(model.xyz_gradient_accum, model.denom, model.max_radii2D)) if folded else None
            pkg = render(cams[it % 3], model, pipe, bg, separate_sh=True, fold=fold)
            loss = training_loss_fused(pkg["render"], gt, 0.2)
            # --- somebody else renders in between ---
            other, got = viewer_grads()
            for a, b in zip(got, plain_viewer):
                assert torch.equal(a, b)                                   # the viewer's backward is an ordinary one
            with torch.no_grad():
                render(cams[2], model, pipe, bg, separate_sh=True)         # forward-only render of the SAME model
            if folded:
                assert not fold.optimizer_taken and not fold.stats_taken   # nobody consumed the request
            # --- now the armed call's backward ---
            loss.backward()
            if folded:
                assert fold.optimizer_taken and fold.stats_taken
                assert all(p.grad is None for p in model.parameters())     # the step rode in the backward: nothing materialised
            else:
                model.add_densification_stats(pkg["viewspace_points"], pkg["visibility_filter"], pkg["radii"])
                opt.step()
                opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        res[folded] = [(p.detach().clone(), opt.state[p]["exp_avg"].clone(), opt.state[p]["exp_avg_sq"].clone())
                       for p in model.parameters()] + [(model.xyz_gradient_accum.clone(), model.denom.clone(), model.max_radii2D.clone())]
    for a, b in zip(res[False], res[True]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    # a retained graph's second backward does not step a second time
    model = GaussianModel.from_raw(make_gaussians(800, 1, seed=195, scale_factor=0.9).to("cuda"))
    opt = model.training_setup(optimizer="hip")
    fold = dgr.BackwardFold(optimizer=opt)
    pkg = render(cams[0], model, pipe, bg, separate_sh=True, fold=fold)
    l = (pkg["render"] * up).sum()
    l.backward(retain_graph=True)
    after_first = [p.detach().clone() for p in model.parameters()]
    assert fold.optimizer_taken and all(p.grad is None for p in model.parameters())
    l.backward()
    torch.cuda.synchronize()
    assert all(torch.equal(a, p.detach()) for a, p in zip(after_first, model.parameters()))
    assert all(p.grad is not None for p in model.parameters() if p.numel())    # the second one is a plain backward


@pytest.mark.parametrize("kind", ["hip", "hip_sparse"])
def test_optimizer_step_folded_into_backward_is_bit_identical(kind):
    """gsr_backward_adam (the Adam / SparseGaussianAdam update applied by the rasterizer backward's last kernel, gradients
    never stored) against backward + the one-launch optimizer kernel: parameters and both moments equal bit for bit after
    several training steps, densification statistics (means2D gradient) included."""
    import diff_gaussian_rasterization as dgr
    from gaussian_renderer import render, PipelineParams
    from scene_utils import make_gaussians, fibonacci_cameras, GaussianModel, Trainer
    cams = fibonacci_cameras(3, 176, 112, seed=91, device="cuda")
    bg = torch.tensor([0.1, 0.2, 0.05], device="cuda")
    pipe = PipelineParams()
    teacher = GaussianModel.from_raw(make_gaussians(3000, 3, seed=92, scale_factor=0.7).to("cuda"), requires_grad=False)
    with torch.no_grad():
        gts = {i: render(c, teacher, pipe, bg)["render"].clone() for i, c in enumerate(cams)}
    runs = {}
    for fused in (False, True):
        model = GaussianModel.from_raw(make_gaussians(3000, 3, seed=93, scale_factor=0.7).to("cuda"))
        tr = Trainer(model, cams, gts, render, pipe, bg, separate_sh=True, optimizer=kind + ("_fused" if fused else ""))
        n0 = dgr.call_stats().get("folded_backwards", 0)
        for it in range(7):
            tr.step(it % 3)
        tr.finish()
        torch.cuda.synchronize()
        assert dgr.call_stats().get("folded_backwards", 0) - n0 == (7 if fused else 0)
        st = {}
        for name, p in zip(("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"), model.parameters()):
            s = tr.optimizer.state[p]
            st[name] = (p.detach().clone(), s["exp_avg"].clone(), s["exp_avg_sq"].clone())
        runs[fused] = (st, model.xyz_gradient_accum.clone(), model.denom.clone())
    for name in runs[False][0]:
        for a, b, what in zip(runs[False][0][name], runs[True][0][name], ("param", "exp_avg", "exp_avg_sq")):
            assert torch.equal(a, b), (name, what, float((a - b).abs().max()))
    assert torch.equal(runs[False][1], runs[True][1]) and torch.equal(runs[False][2], runs[True][2])
    # and the parameters did move
    ref = GaussianModel.from_raw(make_gaussians(3000, 3, seed=93, scale_factor=0.7).to("cuda"))
    assert not torch.equal(ref._features_rest, runs[True][0]["f_rest"][0])


def test_split_dense_adam_culled_rows_on_side_stream_is_bit_identical():
    """hip_fused can split the dense update (Trainer.split_rows): rows without tile instances on a side stream during the compositing kernels
    (gsr_adam_step_culled_rows, launched by the rasterizer's backward), rows with instances in the backward (gsr_backward_adam, sparse = 2).  Against the unsplit
    folded update (Trainer.split_rows off) on a scene where a good part of the Gaussians is off-screen."""
    import diff_gaussian_rasterization as dgr
    from gaussian_renderer import render, PipelineParams
    from scene_utils import make_gaussians, fibonacci_cameras, GaussianModel, Trainer
    cams = fibonacci_cameras(3, 144, 96, seed=191, device="cuda")
    bg = torch.zeros(3, device="cuda")
    pipe = PipelineParams()
    def scene(seed):
        raw = make_gaussians(5000, 3, seed=seed, scale_factor=0.6)
        raw.xyz[::3] *= 3.0                      # a third of the cloud far outside the frustum
        return raw
    teacher = GaussianModel.from_raw(scene(192).to("cuda"), requires_grad=False)
    with torch.no_grad():
        gts = {i: render(c, teacher, pipe, bg)["render"].clone() for i, c in enumerate(cams)}
    out = {}
    for split in (False, True):
        model = GaussianModel.from_raw(scene(193).to("cuda"))
        tr = Trainer(model, cams, gts, render, pipe, bg, separate_sh=True, optimizer="hip_fused")
        tr.split_rows = split
        culled = 0
        for it in range(6):
            o = tr.step(it % 3)
            culled += int((o["radii"] == 0).sum())
        tr.finish()
        torch.cuda.synchronize()
        assert culled > 3000
        out[split] = [(p.detach().clone(), tr.optimizer.state[p]["exp_avg"].clone(), tr.optimizer.state[p]["exp_avg_sq"].clone())
                      for p in model.parameters()]
    for a, b in zip(out[False], out[True]):
        for x, y in zip(a, b):
            assert torch.equal(x, y), float((x - y).abs().max())


@pytest.mark.parametrize("shape,masked", [((1, 67, 45), False), ((1, 270, 481), True), ((1, 2160, 3840), True)])
def test_l1_mean_loss_matches_torch(shape, masked):
    """fused_ssim.l1_mean_loss (the inverse-depth term of reference train.py:124-132) against the torch expression it replaces,
    value and gradient (float64 on the CPU), non-unit upstream gradient, exact ties, bitwise reproducible."""
    from fused_ssim import l1_mean_loss
    gen = torch.Generator().manual_seed(sum(shape))
    a = torch.rand(*shape, generator=gen, dtype=torch.float64)
    b = a + 0.05 * torch.randn(*shape, generator=gen, dtype=torch.float64)
    b[:, :3, :5] = a[:, :3, :5]                               # exact ties: sign(0) = 0 like torch
    mask = (torch.rand(*shape, generator=gen) > 0.3).double() if masked else None
    a_ref = a.clone().requires_grad_(True)
    ref = 0.7 * torch.abs((a_ref - b) * (mask if masked else 1.0)).mean()
    (2.5 * ref).backward()
    runs = []
    for _ in range(2):
        a_gpu = a.float().cuda().requires_grad_(True)
        val = l1_mean_loss(a_gpu, b.float().cuda(), 0.7, mask.float().cuda() if masked else None)
        (2.5 * val).backward()
        runs.append((val.item(), a_gpu.grad.clone()))
    assert abs(runs[0][0] - ref.item()) <= 2e-6 * max(1.0, abs(ref.item()))
    g, g_ref = runs[0][1].cpu().double(), a_ref.grad
    # fp32 rounding of a and b can flip the sign of a difference that is ~1e-8: compare away from ties
    far = (a - b).abs() > 1e-6
    assert torch.allclose(g[far], g_ref[far], rtol=1e-5, atol=1e-12)
    assert bool((g[:, :3, :5] == 0).all())
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])
