"""Sub-results of the hot path pinned to vectors generated from the REFERENCE's own code (tests/golden/reference_cov_loss.npz,
made by tests/golden/make_reference_fixtures_cov_loss.py in the build container):

  (a) the 3-D covariance Sigma = R diag(mod s)^2 R^T and its packing: reference utils/general_utils.py:64-110
      (`strip_symmetric`, `build_rotation`, `build_scaling_rotation`) composed as scene/gaussian_model.py:32-36, and its
      Jacobians with respect to scale and RAW quaternion by autograd through those functions;
  (b) d ssim / d img1, reference utils/loss_utils.py:100-159 by autograd;
  (c) the training loss of train.py:114-121, value and gradient.

CPU half (not gpu): the oracle restatements and the product's host-side `GaussianModel.get_covariance` against the fixture.
GPU half: `cov3d_from_sr` of csrc/preprocess.hip (forward AND the Sigma -> (scale, quaternion) backward of k_preprocess_bwd),
`k_ssim_bwd` and the fused L1 + D-SSIM pair of csrc/ssim.hip.

Tolerances: float64 restatements 1e-12 (relative to max(1, |value|)); fp32 kernels: images |err| <= 2e-5, gradients rel-L2 <= 1e-4
and max-abs <= 1e-4 max|g|, loss values |err| <= 2e-6.
"""
import math
import os

import numpy as np
import pytest
import torch

from oracle import gs_oracle as O
from oracle import loss_oracle
from helpers import rel_l2

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_cov_loss.npz"))
MODS = (1.0, 0.5, 2.3)
LOSS_TAGS = ("s", "m", "one")


def _t(name, dtype=torch.float64):
    return torch.tensor(G[name], dtype=dtype)


def _close(a, b, rel):
    return bool(((a - b).abs() <= rel * torch.clamp(b.abs(), min=1.0)).all())


# ----------------------------------------------------------------------------------------------------------------------------
# CPU: oracle and host logic against the reference's vectors
# ----------------------------------------------------------------------------------------------------------------------------
def test_fixture_is_what_the_generator_says():
    assert G["cov6_f64"].shape == (300, 6) and G["cov6_dscale"].shape == (300, 6, 3) and G["cov6_dquat_raw"].shape == (300, 6, 4)
    assert np.allclose(G["cov6_f32"], G["cov6_f64"], rtol=2e-5, atol=1e-9)     # the as-written float32 run and the promoted one agree
    assert np.allclose(np.linalg.norm(G["cov_quat_unit"], axis=1), 1.0, atol=1e-14)


def test_oracle_cov3d_matches_reference_build_covariance():
    s, qu, mod = _t("cov_scales"), _t("cov_quat_unit"), G["cov_modifier"]
    for m in MODS:
        sel = torch.tensor(mod == m)
        c = O.cov3d_from_scale_rot(s[sel], qu[sel], m)
        assert _close(c, _t("cov6_f64")[sel], 1e-12), float((c - _t("cov6_f64")[sel]).abs().max())
        c32 = O.cov3d_from_scale_rot(s[sel].float(), qu[sel].float(), m)
        ref32 = _t("cov6_f32", torch.float32)[sel]
        assert bool(((c32 - ref32).abs() <= 3e-6 * torch.clamp(ref32.abs().amax(dim=1, keepdim=True), min=1e-12)).all())


def test_product_get_covariance_matches_reference():
    """scene_utils.model.GaussianModel.get_covariance is the host-side mirror of scene/gaussian_model.py:32-36 (raw rotation in,
    build_rotation normalises)."""
    from scene_utils.model import GaussianModel
    s, q, mod = _t("cov_scales"), _t("cov_quat_raw"), G["cov_modifier"]
    m = GaussianModel(0)
    for mv in MODS:
        sel = torch.tensor(mod == mv)
        m._scaling, m._rotation = torch.log(s[sel]), q[sel]
        c = m.get_covariance(mv)
        assert _close(c, _t("cov6_f64")[sel], 1e-12)


def test_oracle_cov3d_jacobians_match_reference_autograd():
    s, q, mod = _t("cov_scales"), _t("cov_quat_raw"), G["cov_modifier"]
    for m in MODS:
        sel = torch.nonzero(torch.tensor(mod == m)).flatten()
        ss = s[sel].clone().requires_grad_(True)
        qq = q[sel].clone().requires_grad_(True)
        c = O.cov3d_from_scale_rot(ss, torch.nn.functional.normalize(qq), m)
        for k in range(6):
            gs, gq = torch.autograd.grad(c[:, k].sum(), (ss, qq), retain_graph=True)
            assert _close(gs, _t("cov6_dscale")[sel, k], 1e-11)
            assert _close(gq, _t("cov6_dquat_raw")[sel, k], 1e-11)


@pytest.mark.parametrize("tag", LOSS_TAGS)
def test_oracle_ssim_and_training_loss_gradients_match_reference(tag):
    a, b = _t(f"loss_{tag}_a"), _t(f"loss_{tag}_b")
    x = a.clone().requires_grad_(True)
    v = loss_oracle.ssim(x, b)
    (g,) = torch.autograd.grad(v, x)
    assert math.isclose(v.item(), float(G[f"loss_{tag}_ssim_f64"]), rel_tol=1e-13)
    assert (g - _t(f"loss_{tag}_dssim_da_f64")).abs().max().item() <= 1e-15 + 1e-12 * g.abs().max().item()
    for lam in (0.2, 0.5):
        x = a.clone().requires_grad_(True)
        loss = loss_oracle.training_loss(x, b, lam)
        (g,) = torch.autograd.grad(loss, x)
        assert math.isclose(loss.item(), float(G[f"loss_{tag}_train_l{int(lam * 10)}_f64"]), rel_tol=1e-13)
        assert (g - _t(f"loss_{tag}_dtrain_da_l{int(lam * 10)}_f64")).abs().max().item() <= 1e-12 * g.abs().max().item()
    v32 = loss_oracle.ssim(a.float(), b.float()).item()
    assert abs(v32 - float(G[f"loss_{tag}_ssim_f32"])) <= 1e-6


# ----------------------------------------------------------------------------------------------------------------------------
# GPU: the kernels against the same vectors
# ----------------------------------------------------------------------------------------------------------------------------
W_IMG, H_IMG = 160, 112


def _cov_scene(m):
    """The fixture's Gaussians of modifier m (100 of them), placed on a seeded cloud in front of one camera."""
    from scene_utils import look_at_camera
    sel = torch.nonzero(torch.tensor(G["cov_modifier"] == m)).flatten()
    n = sel.numel()
    gen = torch.Generator().manual_seed(1000 + int(m * 10))
    xyz = (torch.rand(n, 3, generator=gen) * 2.0 - 1.0) * torch.tensor([1.2, 0.8, 0.8])
    opac = 0.2 + 0.75 * torch.rand(n, 1, generator=gen)
    col = torch.rand(n, 3, generator=gen)
    cam = look_at_camera(np.array([0.3, -4.0, 0.5]), np.zeros(3), (0.0, 0.0, 1.0), 0.7, W_IMG, H_IMG)
    return sel, xyz, opac, col, cam


def _settings(cam, m, dev, cls):
    return cls(image_height=H_IMG, image_width=W_IMG, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
               bg=torch.tensor([0.1, 0.3, 0.2]).to(dev), scale_modifier=float(m), viewmatrix=cam.world_view_transform.to(dev),
               projmatrix=cam.full_proj_transform.to(dev), sh_degree=0, campos=cam.camera_center.to(dev), prefiltered=False,
               debug=False, antialiasing=False)


@pytest.mark.gpu
@pytest.mark.parametrize("m", MODS)
def test_kernel_cov3d_from_scale_rotation_matches_reference_covariance(m):
    """Forward: the image the kernels compose from (scales, unit quaternions, modifier) equals the image they compose from the
    REFERENCE's Sigma handed over as cov3D_precomp - identical radii, |colour / inverse depth difference| <= 2e-5 - and both equal
    the oracle's image of the reference's Sigma.  That pins `cov3d_from_sr` (csrc/preprocess.hip) to the reference's arithmetic."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = "cuda"
    sel, xyz, opac, col, cam = _cov_scene(m)
    n = sel.numel()
    rast = GaussianRasterizer(_settings(cam, m, dev, GaussianRasterizationSettings))
    common = dict(means3D=xyz.to(dev), means2D=torch.zeros(n, 3, device=dev), opacities=opac.to(dev), colors_precomp=col.to(dev))
    s32 = _t("cov_scales", torch.float32)[sel].to(dev)
    q32 = _t("cov_quat_unit", torch.float32)[sel].to(dev)
    cov_ref = _t("cov6_f32", torch.float32)[sel].to(dev)
    with torch.no_grad():
        c_sr, r_sr, d_sr = rast(scales=s32, rotations=q32, **common)
        c_pc, r_pc, d_pc = rast(cov3D_precomp=cov_ref, **common)
    torch.cuda.synchronize()
    assert int((r_sr > 0).sum()) >= 60                   # the scene does put most of them on the screen
    assert torch.equal(r_sr, r_pc)
    assert (c_sr - c_pc).abs().max().item() <= 2e-5
    assert (d_sr - d_pc).abs().max().item() <= 2e-5 * max(1.0, d_pc.abs().max().item())
    # and the oracle, fed the reference's float64 Sigma
    so = _settings(cam, m, "cpu", O.OracleSettings)
    co, ro, do = O.rasterize(xyz.double(), torch.zeros(n, 3, dtype=torch.float64), opac.double(), so,
                             colors_precomp=col.double(), cov3D_precomp=_t("cov6_f64")[sel])
    assert torch.equal(ro.to(torch.int32), r_sr.cpu())
    assert (co - c_sr.cpu().double()).abs().max().item() <= 2e-5
    assert (do - d_sr.cpu().double()).abs().max().item() <= 2e-5 * max(1.0, do.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("m", MODS)
def test_kernel_covariance_backward_matches_reference_jacobians(m):
    """Backward: dL/dSigma as the kernels return it for the cov3D_precomp call form, contracted with the REFERENCE's autograd
    Jacobians d cov6 / d scale and d cov6 / d raw-quaternion, equals the dL/dscale and dL/d(raw quaternion) the kernels return
    for the (scales, rotations) call form (k_preprocess_bwd's Sigma -> (s, q) chain, SURVEY A.7 v; the normalisation is torch's
    F.normalize in both worlds, as scene/gaussian_model.py:46).  The modifier's chain factor follows autograd (DESIGN q3)."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = "cuda"
    sel, xyz, opac, col, cam = _cov_scene(m)
    n = sel.numel()
    rast = GaussianRasterizer(_settings(cam, m, dev, GaussianRasterizationSettings))
    gen = torch.Generator().manual_seed(5)
    gc = torch.randn(3, H_IMG, W_IMG, generator=gen).to(dev)
    gd = torch.randn(1, H_IMG, W_IMG, generator=gen).to(dev)
    common = dict(means3D=xyz.to(dev), opacities=opac.to(dev), colors_precomp=col.to(dev))

    cov = _t("cov6_f32", torch.float32)[sel].to(dev).requires_grad_(True)
    c, r, d = rast(means2D=torch.zeros(n, 3, device=dev, requires_grad=True), cov3D_precomp=cov, **common)
    ((c * gc).sum() + (d * gd).sum()).backward()
    g_cov = cov.grad.double().cpu()

    s = _t("cov_scales", torch.float32)[sel].to(dev).requires_grad_(True)
    q_raw = _t("cov_quat_raw", torch.float32)[sel].to(dev).requires_grad_(True)
    c2, r2, d2 = rast(means2D=torch.zeros(n, 3, device=dev, requires_grad=True), scales=s,
                      rotations=torch.nn.functional.normalize(q_raw), **common)
    ((c2 * gc).sum() + (d2 * gd).sum()).backward()
    torch.cuda.synchronize()
    assert torch.equal(r, r2)
    vis = (r > 0).cpu()
    assert int(vis.sum()) >= 60 and float(g_cov[vis].abs().sum()) > 0
    exp_s = torch.einsum("nk,nkj->nj", g_cov, _t("cov6_dscale")[sel])
    exp_q = torch.einsum("nk,nkj->nj", g_cov, _t("cov6_dquat_raw")[sel])
    for got, exp, name in ((s.grad, exp_s, "scale"), (q_raw.grad, exp_q, "quaternion")):
        got = got.double().cpu()
        assert rel_l2(got, exp) <= 1e-4, (name, rel_l2(got, exp))
        assert (got - exp).abs().max().item() <= 1e-4 * exp.abs().max().item(), name
        assert float(got[~vis].abs().sum()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("tag", LOSS_TAGS)
def test_kernel_ssim_gradient_matches_reference_autograd(tag):
    from diff_gaussian_rasterization._C import fusedssim, fusedssim_backward
    from fused_ssim import fused_ssim
    a = _t(f"loss_{tag}_a", torch.float32).cuda()
    b = _t(f"loss_{tag}_b", torch.float32).cuda()
    exp = _t(f"loss_{tag}_dssim_da_f64")
    # the `_C.fusedssim` / `_C.fusedssim_backward` interface of reference utils/loss_utils.py:16-38
    m = fusedssim(0.01 ** 2, 0.03 ** 2, a, b)
    assert abs(m.mean().item() - float(G[f"loss_{tag}_ssim_f64"])) <= 2e-6
    g = fusedssim_backward(0.01 ** 2, 0.03 ** 2, a, b, torch.full_like(a, 1.0 / a.numel())).double().cpu()
    assert rel_l2(g, exp) <= 1e-4, rel_l2(g, exp)
    assert (g - exp).abs().max().item() <= 1e-4 * exp.abs().max().item()
    # the `fused_ssim` module's autograd form (train.py:116-117: image.unsqueeze(0))
    x = a.clone().requires_grad_(True)
    v = fused_ssim(x.unsqueeze(0), b.unsqueeze(0))
    v.backward()
    assert abs(v.item() - float(G[f"loss_{tag}_ssim_f64"])) <= 2e-6
    g2 = x.grad.double().cpu()
    assert rel_l2(g2, exp) <= 1e-4 and (g2 - exp).abs().max().item() <= 1e-4 * exp.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("lam", [0.2, 0.5])
@pytest.mark.parametrize("tag", LOSS_TAGS)
def test_kernel_training_loss_matches_reference(tag, lam):
    """(1 - l) l1_loss + l (1 - ssim), train.py:114-121: value and gradient of the fused HIP pair against the reference's."""
    from scene_utils.losses import training_loss_fused
    a = _t(f"loss_{tag}_a", torch.float32).cuda().requires_grad_(True)
    b = _t(f"loss_{tag}_b", torch.float32).cuda()
    loss = training_loss_fused(a, b, lam)
    loss.backward()
    k = int(lam * 10)
    assert abs(loss.item() - float(G[f"loss_{tag}_train_l{k}_f64"])) <= 2e-6
    exp = _t(f"loss_{tag}_dtrain_da_l{k}_f64")
    g = a.grad.double().cpu()
    assert rel_l2(g, exp) <= 1e-4, rel_l2(g, exp)
    assert (g - exp).abs().max().item() <= 1e-4 * exp.abs().max().item()
