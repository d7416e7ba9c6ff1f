"""Deterministic synthetic scenes for the five BASELINE.json configs.

Gaussians are generated on the CPU with a seeded generator and then moved to the device, so the CPU
oracle and the HIP path see identical tensors.  Parameter shapes / activations follow the reference's
`GaussianModel` (`scene/gaussian_model.py:38-46,101-124,147-153`): raw log-scales, raw quaternions, raw
(logit) opacities, `features_dc [P,1,3]`, `features_rest [P,M-1,3]`.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

from .cameras import fibonacci_cameras

SH_C0 = 0.28209479177387814

# cfg id -> (P, sh degree, W, H, views, antialiasing, depth-grad)
CONFIGS = {
    1: dict(P=10_000, deg=0, W=256, H=256, views=1, antialiasing=False, depth_grad=False),
    2: dict(P=100_000, deg=3, W=1920, H=1080, views=1, antialiasing=False, depth_grad=False),
    3: dict(P=1_000_000, deg=3, W=1920, H=1080, views=100, antialiasing=False, depth_grad=False),
    4: dict(P=5_000_000, deg=3, W=3840, H=2160, views=1, antialiasing=True, depth_grad=True),
    5: dict(P=50_000, deg=3, W=1280, H=720, views=100, antialiasing=False, depth_grad=False, P_final=500_000),
}


@dataclass
class RawGaussians:
    """Leaf parameters in the reference model's raw (pre-activation) space."""
    xyz: torch.Tensor            # [P,3]
    features_dc: torch.Tensor    # [P,1,3]
    features_rest: torch.Tensor  # [P,M-1,3]
    scaling: torch.Tensor        # [P,3] log-scale
    rotation: torch.Tensor       # [P,4] raw quaternion (w,x,y,z)
    opacity: torch.Tensor        # [P,1] logit
    sh_degree: int

    def to(self, device):
        return RawGaussians(*(t.to(device) for t in self.tensors()), self.sh_degree)

    def tensors(self):
        return (self.xyz, self.features_dc, self.features_rest, self.scaling, self.rotation, self.opacity)

    def names(self):
        return ("xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity")

    def activated(self):
        """What `render()` passes to the rasterizer (reference gaussian_renderer/__init__.py:54-86)."""
        return dict(
            means3D=self.xyz,
            opacities=torch.sigmoid(self.opacity),
            scales=torch.exp(self.scaling),
            rotations=torch.nn.functional.normalize(self.rotation),
            shs=torch.cat((self.features_dc, self.features_rest), dim=1),
        )


def make_gaussians(P: int, deg: int, seed: int, scale_factor: float = 0.25, box: float = 1.3) -> RawGaussians:
    gen = torch.Generator(device="cpu").manual_seed(int(seed))
    xyz = torch.rand(P, 3, generator=gen) * (2 * box) - box
    mu_s = math.log(scale_factor * ((2 * box) ** 3 / P) ** (1.0 / 3.0))
    log_s = mu_s + 0.4 * torch.randn(P, 3, generator=gen)
    quat = torch.randn(P, 4, generator=gen)
    opac = 1.5 * torch.randn(P, 1, generator=gen)
    f_dc = (torch.rand(P, 1, 3, generator=gen) - 0.5) / SH_C0
    f_rest = 0.05 * torch.randn(P, (deg + 1) ** 2 - 1, 3, generator=gen)
    return RawGaussians(xyz, f_dc, f_rest, log_s, quat, opac, deg)


def make_config(cfg: int, device="cpu", P: int | None = None, views: int | None = None,
                W: int | None = None, H: int | None = None, splat_scale: float = 1.0):
    """-> (RawGaussians, [MiniCam], config dict).  Seeds: 1000*cfg (Gaussians), 1000*cfg+1 (cameras).
    splat_scale: multiplier on the recipe's splat size (c = 0.25); 2.0 gives a compositing-heavy variant of the
    same scene (tile instances per Gaussian ~ x3.5), not a BASELINE config."""
    c = dict(CONFIGS[cfg])
    if P is not None:
        c["P"] = P
    if views is not None:
        c["views"] = views
    if W is not None:
        c["W"] = W
    if H is not None:
        c["H"] = H
    g = make_gaussians(c["P"], c["deg"], 1000 * cfg, scale_factor=0.25 * splat_scale).to(device)
    c["splat_scale"] = float(splat_scale)
    cams = fibonacci_cameras(c["views"], c["W"], c["H"], seed=1000 * cfg + 1, device=device)
    return g, cams, c
