"""Generates tests/golden/reference_helpers.npz  (run ONCE in the build container; never on the GPU box).

Imports the few reference helpers that are importable on CPU and records their
outputs on seeded inputs.  Only DATA (inputs + expected outputs) is written; no reference source travels.

  python tests/golden/make_reference_fixtures.py        # needs /root/reference

Pins:  SH->RGB (utils/sh_utils.py:57-112), getWorld2View2 / getProjectionMatrix / fov2focal / focal2fov
(utils/graphics_utils.py:38-77), MiniCam.camera_center (scene/cameras.py:74-85, loaded by file path because
scene/__init__.py cannot be imported - SURVEY §0.3), psnr (utils/image_utils.py:17-19),
l1_loss / ssim (utils/loss_utils.py:40-52,100-159), RGB2SH (utils/sh_utils.py:114-115).
"""
import importlib.util
import math
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_helpers.npz")


def main():
    sys.path.insert(0, REF)
    from utils.sh_utils import eval_sh, RGB2SH                       # noqa: E402
    from utils.graphics_utils import getWorld2View2, getProjectionMatrix, fov2focal, focal2fov  # noqa: E402
    from utils.image_utils import psnr                               # noqa: E402
    from utils.loss_utils import l1_loss, ssim                       # noqa: E402
    spec = importlib.util.spec_from_file_location("ref_cameras", os.path.join(REF, "scene", "cameras.py"))
    ref_cameras = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_cameras)

    gen = torch.Generator().manual_seed(20241220)
    out = {}

    # --- SH -> RGB, degrees 0..3, float64 inputs so the fixture is the "true" value -----------------
    P = 257
    sh = torch.randn(P, 16, 3, generator=gen, dtype=torch.float64)
    xyz = torch.randn(P, 3, generator=gen, dtype=torch.float64) * 2.0
    campos = torch.tensor([0.3, -1.2, 2.5], dtype=torch.float64)
    dirs = xyz - campos[None]
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    out["sh_coeffs"] = sh.numpy()
    out["sh_xyz"] = xyz.numpy()
    out["sh_campos"] = campos.numpy()
    for deg in range(4):
        # reference layout: [P, 3, K]  (gaussian_renderer/__init__.py:75)
        rgb = eval_sh(deg, sh.transpose(1, 2), dirs)
        out[f"sh_rgb_deg{deg}"] = torch.clamp_min(rgb + 0.5, 0.0).numpy()   # __init__.py:79
        out[f"sh_raw_deg{deg}"] = rgb.numpy()
    out["rgb2sh_in"] = np.linspace(0, 1, 11)
    out["rgb2sh_out"] = RGB2SH(torch.linspace(0, 1, 11, dtype=torch.float64)).numpy()

    # --- camera math -------------------------------------------------------------------------------
    n_cam = 6
    Rs, Ts, w2v, projs, fulls, centers, fovs = [], [], [], [], [], [], []
    for i in range(n_cam):
        q = torch.randn(4, generator=gen, dtype=torch.float64)
        q = (q / q.norm()).numpy()
        w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        T = torch.randn(3, generator=gen, dtype=torch.float64).numpy() * 3.0
        fovx = 0.4 + 0.2 * i
        width, height = 640 + 32 * i, 360 + 16 * i
        fovy = focal2fov(fov2focal(fovx, width), height)
        wv = torch.tensor(getWorld2View2(R, T)).transpose(0, 1)
        proj = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
        full = wv.unsqueeze(0).bmm(proj.unsqueeze(0)).squeeze(0)
        cam = ref_cameras.MiniCam(width, height, fovy, fovx, 0.01, 100.0, wv, full)
        Rs.append(R); Ts.append(T); w2v.append(wv.numpy()); projs.append(proj.numpy())
        fulls.append(full.numpy()); centers.append(cam.camera_center.numpy())
        fovs.append([fovx, fovy, width, height, fov2focal(fovx, width)])
    out["cam_R"] = np.stack(Rs); out["cam_T"] = np.stack(Ts)
    out["cam_world_view"] = np.stack(w2v); out["cam_proj"] = np.stack(projs)
    out["cam_full_proj"] = np.stack(fulls); out["cam_center"] = np.stack(centers)
    out["cam_fov"] = np.array(fovs, dtype=np.float64)
    # translate/scale variant (graphics_utils.py:38)
    out["cam_w2v_ts"] = getWorld2View2(Rs[0], Ts[0], np.array([0.1, -0.2, 0.3]), 1.5)

    # --- metrics ------------------------------------------------------------------------------------
    a = torch.rand(3, 17, 23, generator=gen)
    b = (a + 0.05 * torch.randn(3, 17, 23, generator=gen)).clamp(0, 1)
    out["img_a"] = a.numpy(); out["img_b"] = b.numpy()
    out["psnr_ab"] = psnr(a, b).numpy()
    out["l1_ab"] = np.array(l1_loss(a, b).item())
    a2 = torch.rand(3, 40, 52, generator=gen)
    b2 = (a2 + 0.1 * torch.randn(3, 40, 52, generator=gen)).clamp(0, 1)
    out["img_a2"] = a2.numpy(); out["img_b2"] = b2.numpy()
    out["ssim_ab2"] = np.array(ssim(a2, b2).item())                  # utils/loss_utils.py:100-159

    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
