"""Minimal host-side Gaussian model: the attribute surface `render()` reads from the reference's
`GaussianModel` (reference scene/gaussian_model.py:38-46 activations, :101-124 getters, :147-153 parameter
shapes; `get_features_dc/get_features_rest/get_exposure_from_name` are called by reference
gaussian_renderer/__init__.py:82,113 but missing from the reference model - SURVEY.md 0.3 - and are provided
here).  Densification / PLY IO are out of the hot path and not part of this class.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .synthetic import RawGaussians


def _build_rotation(r):
    """reference utils/general_utils.py:78-99 (device-agnostic restatement)."""
    q = r / r.norm(dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.view(-1, 3, 3)


class GaussianModel:
    def __init__(self, sh_degree: int):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        self._xyz = self._features_dc = self._features_rest = None
        self._scaling = self._rotation = self._opacity = None
        self._exposure = None
        self.exposure_mapping = {}

    @classmethod
    def from_raw(cls, raw: RawGaussians, requires_grad: bool = True, active_sh_degree=None):
        m = cls(raw.sh_degree)
        m.active_sh_degree = raw.sh_degree if active_sh_degree is None else active_sh_degree
        mk = (lambda t: nn.Parameter(t.clone().contiguous(), requires_grad=True)) if requires_grad else \
            (lambda t: t.clone().contiguous())
        m._xyz, m._features_dc, m._features_rest = mk(raw.xyz), mk(raw.features_dc), mk(raw.features_rest)
        m._scaling, m._rotation, m._opacity = mk(raw.scaling), mk(raw.rotation), mk(raw.opacity)
        return m

    def parameters(self):
        return [self._xyz, self._features_dc, self._features_rest, self._opacity, self._scaling, self._rotation]

    def param_groups(self, position_lr=0.00016, feature_lr=0.0025, opacity_lr=0.025, scaling_lr=0.005,
                     rotation_lr=0.001):
        """Adam groups and learning rates of reference scene/gaussian_model.py:160-168 /
        arguments/__init__.py:76-83 (spatial_lr_scale = 1)."""
        return [
            {"params": [self._xyz], "lr": position_lr, "name": "xyz"},
            {"params": [self._features_dc], "lr": feature_lr, "name": "f_dc"},
            {"params": [self._features_rest], "lr": feature_lr / 20.0, "name": "f_rest"},
            {"params": [self._opacity], "lr": opacity_lr, "name": "opacity"},
            {"params": [self._scaling], "lr": scaling_lr, "name": "scaling"},
            {"params": [self._rotation], "lr": rotation_lr, "name": "rotation"},
        ]

    # -- getters (reference scene/gaussian_model.py:101-124) --
    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._rotation)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_features_dc(self):
        return self._features_dc

    @property
    def get_features_rest(self):
        return self._features_rest

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    def get_covariance(self, scaling_modifier=1):
        """reference scene/gaussian_model.py:32-36 (note: passes the RAW rotation; build_rotation normalises)."""
        R = _build_rotation(self._rotation)
        L = R * (scaling_modifier * self.get_scaling)[:, None, :]
        S = L @ L.transpose(1, 2)
        return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=1)

    def get_exposure_from_name(self, image_name):
        if self._exposure is None:
            return torch.eye(3, 4, device=self._xyz.device)
        return self._exposure[self.exposure_mapping[image_name]]

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1
