"""Camera math for the rasterizer boundary.

Restates (does not import) the conventions of the reference:
  * `utils/graphics_utils.py:38-49`  getWorld2View2     (R is stored transposed, T = w2c translation)
  * `utils/graphics_utils.py:51-71`  getProjectionMatrix (z_sign = +1, w = z_view)
  * `utils/graphics_utils.py:73-77`  fov2focal / focal2fov
  * `scene/cameras.py:63-72,74-85`   world_view_transform = W2C^T, full_proj_transform = W2C^T . P^T,
                                     camera_center = inverse(world_view_transform)[3,:3]
Pinned against the reference's own functions by tests/golden/reference_helpers.npz.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def fov2focal(fov: float, pixels: int) -> float:
    return pixels / (2.0 * math.tan(fov / 2.0))


def focal2fov(focal: float, pixels: int) -> float:
    return 2.0 * math.atan(pixels / (2.0 * focal))


def world_to_view(R: np.ndarray, t: np.ndarray, translate=(0.0, 0.0, 0.0), scale: float = 1.0) -> np.ndarray:
    Rt = np.zeros((4, 4), dtype=np.float64)
    Rt[:3, :3] = np.asarray(R, dtype=np.float64).T
    Rt[:3, 3] = np.asarray(t, dtype=np.float64)
    Rt[3, 3] = 1.0
    c2w = np.linalg.inv(Rt)
    c2w[:3, 3] = (c2w[:3, 3] + np.asarray(translate, dtype=np.float64)) * scale
    return np.linalg.inv(c2w).astype(np.float32)


def projection_matrix(znear: float, zfar: float, fovX: float, fovY: float) -> torch.Tensor:
    ty, tx = math.tan(fovY / 2.0), math.tan(fovX / 2.0)
    top, right = ty * znear, tx * znear
    bottom, left = -top, -right
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


class MiniCam:
    """Same attribute surface as reference `scene/cameras.py:74-85` (what `render()` reads:
    `gaussian_renderer/__init__.py:33-46`)."""

    def __init__(self, width, height, fovy, fovx, znear, zfar, world_view_transform, full_proj_transform,
                 image_name: str = ""):
        self.image_width = int(width)
        self.image_height = int(height)
        self.FoVy = float(fovy)
        self.FoVx = float(fovx)
        self.znear = znear
        self.zfar = zfar
        # contiguous once, here: the rasterizer wrapper hands raw pointers to the library and would otherwise copy the
        # (transposed-view) matrices on every call
        self.world_view_transform = world_view_transform.contiguous()
        self.full_proj_transform = full_proj_transform.contiguous()
        self.camera_center = torch.inverse(world_view_transform.float().cpu())[3][:3].to(world_view_transform.device)
        self.image_name = image_name

    def to(self, device):
        self.world_view_transform = self.world_view_transform.to(device).contiguous()
        self.full_proj_transform = self.full_proj_transform.to(device).contiguous()
        self.camera_center = self.camera_center.to(device)
        return self


def camera_from_RT(R: np.ndarray, T: np.ndarray, fovx: float, fovy: float, width: int, height: int,
                   znear: float = 0.01, zfar: float = 100.0, device="cpu", name: str = "") -> MiniCam:
    wv = torch.tensor(world_to_view(R, T)).transpose(0, 1)
    proj = projection_matrix(znear, zfar, fovx, fovy).transpose(0, 1)
    full = wv.unsqueeze(0).bmm(proj.unsqueeze(0)).squeeze(0)
    return MiniCam(width, height, fovy, fovx, znear, zfar, wv.to(device), full.to(device), name)


def look_at_camera(eye, target, up, fovx: float, width: int, height: int, device="cpu", name: str = "") -> MiniCam:
    """COLMAP-style camera (x right, y down, z forward) at `eye` looking at `target`."""
    eye = np.asarray(eye, dtype=np.float64)
    f = np.asarray(target, dtype=np.float64) - eye
    f /= np.linalg.norm(f)
    upv = np.asarray(up, dtype=np.float64)
    if abs(np.dot(f, upv)) > 0.999:                          # looking along `up`: pick another
        upv = np.array([0.0, 1.0, 0.0]) if abs(upv[1]) < 0.9 else np.array([1.0, 0.0, 0.0])
    r = np.cross(f, upv)
    r /= np.linalg.norm(r)
    d = np.cross(f, r)                                       # camera "down"
    c2w = np.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = r, d, f, eye
    w2c = np.linalg.inv(c2w)
    R = w2c[:3, :3].T                                        # stored transposed (dataset_readers.py:208-209)
    T = w2c[:3, 3]
    fovy = focal2fov(fov2focal(fovx, width), height)         # dataset_readers.py:224
    return camera_from_RT(R, T, fovx, fovy, width, height, device=device, name=name)


def fibonacci_cameras(n_views: int, width: int, height: int, radius: float = 4.0, fovx: float = 0.6911,
                      seed: int = 0, device="cpu"):
    """V cameras on a Fibonacci sphere looking at the origin, up = +z."""
    rng = np.random.default_rng(seed)
    phase = rng.uniform(0.0, 2.0 * math.pi)
    golden = math.pi * (3.0 - math.sqrt(5.0))
    cams = []
    for i in range(n_views):
        z = 1.0 - 2.0 * (i + 0.5) / n_views
        rho = math.sqrt(max(0.0, 1.0 - z * z))
        th = phase + golden * i
        eye = radius * np.array([rho * math.cos(th), rho * math.sin(th), z])
        cams.append(look_at_camera(eye, (0.0, 0.0, 0.0), (0.0, 0.0, 1.0), fovx, width, height,
                                   device=device, name=f"view_{i:04d}"))
    return cams
