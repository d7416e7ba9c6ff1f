"""End-to-end training parity (north_star: "PSNR within 0.1 dB of reference"): the same scene, the same views, the same
initialisation trained (a) on the MI355X with the product stack - HIP rasterizer, fused L1+D-SSIM loss, one-launch HIP Adam -
and (b) on the CPU with the test stack - oracle rasterizer (autograd gradients), pure-PyTorch loss, torch.optim.Adam."""
import math

import pytest
import torch

from oracle.loss_oracle import training_loss

from scene_utils import make_gaussians, fibonacci_cameras, GaussianModel, Trainer, psnr

pytestmark = pytest.mark.gpu


def _oracle_render(cam, pc, pipe, bg, separate_sh=False, **kw):
    from oracle import gs_oracle as O
    m2d = torch.zeros_like(pc.get_xyz, requires_grad=True) + 0
    if m2d.requires_grad:
        m2d.retain_grad()
    s = O.OracleSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg, 1.0,
                         cam.world_view_transform, cam.full_proj_transform, pc.active_sh_degree, cam.camera_center, False,
                         False, False)
    color, radii, invd = O.rasterize(pc.get_xyz, m2d, pc.get_opacity, s, shs=pc.get_features, scales=pc.get_scaling,
                                     rotations=pc.get_rotation)
    return {"render": color, "viewspace_points": m2d, "visibility_filter": radii > 0, "radii": radii, "depth": invd}


def test_training_psnr_matches_oracle_training():
    from gaussian_renderer import render, PipelineParams
    P, deg, W, H, V, iters = 1500, 2, 64, 48, 4, 40
    init = make_gaussians(P, deg, seed=40, scale_factor=0.9)
    teacher = make_gaussians(P, deg, seed=41, scale_factor=0.9)
    pipe = PipelineParams()
    # ground truth from the oracle (CPU), shared by both runs
    cams_cpu = fibonacci_cameras(V, W, H, seed=42)
    tmodel = GaussianModel.from_raw(teacher, requires_grad=False)
    bg = torch.zeros(3)
    with torch.no_grad():
        gts = {i: _oracle_render(c, tmodel, pipe, bg)["render"].clamp(0, 1) for i, c in enumerate(cams_cpu)}

    # (b) CPU test stack
    m_cpu = GaussianModel.from_raw(init)
    t_cpu = Trainer(m_cpu, cams_cpu, gts, _oracle_render, pipe, bg, optimizer="torch", loss=training_loss)
    for it in range(iters):
        t_cpu.step(it % V)
    # (a) product stack on the GPU
    cams_gpu = fibonacci_cameras(V, W, H, seed=42, device="cuda")
    m_gpu = GaussianModel.from_raw(init.to("cuda"))
    t_gpu = Trainer(m_gpu, cams_gpu, {i: g.cuda() for i, g in gts.items()}, render, pipe, bg.cuda(), optimizer="hip",
                    loss="hip", separate_sh=True)
    losses = []
    for it in range(iters):
        losses.append(float(t_gpu.step(it % V)["loss"]))
    t_gpu.finish()      # the last step's SH update may still be waiting for "the next forward" (Trainer overlap schedule)
    assert losses[-1] < losses[0]                                   # it actually trains

    with torch.no_grad():
        for i in range(V):
            img_cpu = _oracle_render(cams_cpu[i], m_cpu, pipe, bg)["render"]
            img_gpu = render(cams_gpu[i], m_gpu, pipe, bg.cuda())["render"].cpu()
            p_cpu = float(psnr(img_cpu, gts[i]).mean())
            p_gpu = float(psnr(img_gpu, gts[i]).mean())
            assert abs(p_cpu - p_gpu) <= 0.1, (i, p_cpu, p_gpu)      # north-star bar
            assert float(psnr(img_gpu, img_cpu).mean()) >= 45.0      # and the two trained models render alike
    # Parameters agree except where Adam (eps = 1e-15, the reference's setting) turns gradient noise of ~1e-10 on barely
    # visible Gaussians into +-lr steps - inherent to that optimizer setting, so the check is statistical.
    for a, b in zip(m_cpu.parameters(), m_gpu.parameters()):
        d = (a.detach() - b.detach().cpu()).abs().flatten()
        assert float(d.median()) <= 1e-4 * max(1.0, float(a.detach().abs().max()))
        assert float((d <= 2e-3 * max(1.0, float(a.detach().abs().max()))).float().mean()) >= 0.9
