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

    def get_raw_geometry(self):
        """(_scaling, _rotation, _opacity): the raw parameters, for a rasterizer that applies the activations itself
        (`GaussianRasterizer.forward(..., raw_activations=True)`); render() prefers this on the HIP device.  None-returning
        on CPU tensors is not needed: render() is only ever called with device models."""
        return self._scaling, self._rotation, self._opacity

    def get_activated(self):
        """(scaling, rotation, opacity) as get_scaling / get_rotation / get_opacity return them, through ONE fused HIP
        launch (and one in the backward) when the parameters live on the device; render() prefers this when the model
        offers it.  CPU tensors (host-logic tests) take the plain torch getters."""
        if self._scaling.is_cuda:
            from .activations import gaussian_activations
            return gaussian_activations(self._scaling, self._rotation, self._opacity)
        return self.get_scaling, self.get_rotation, self.get_opacity

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


# ---------------------------------------------------------------------------------------------------------------------
# Training-time state and densification (reference scene/gaussian_model.py:155-191 training_setup, :226-229 reset_opacity,
# :274-433 optimizer surgery / densify / prune / statistics).  The decisions and the row gather run in HIP
# (csrc/densify.hip) behind the C ABI; there is no PyTorch fallback.
# ---------------------------------------------------------------------------------------------------------------------
def _inverse_sigmoid(x):
    return torch.log(x / (1 - x))


_PARAM_ATTRS = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
_GROUP_NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")


def training_setup(self, optimizer="hip", percent_dense=0.01, **lrs):
    """optimizer: "hip" (FusedAdam), "hip_sparse" (SparseGaussianAdam) or "torch" (torch.optim.Adam, CPU tests)."""
    self.percent_dense = percent_dense
    dev = self._xyz.device
    P = self._xyz.shape[0]
    self.xyz_gradient_accum = torch.zeros((P, 1), device=dev)
    self.denom = torch.zeros((P, 1), device=dev)
    self.max_radii2D = torch.zeros((P,), device=dev)
    groups = self.param_groups(**lrs)
    if optimizer == "torch":
        self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
    elif optimizer == "hip":
        from diff_gaussian_rasterization import FusedAdam
        self.optimizer = FusedAdam(groups, lr=0.0, eps=1e-15)
    elif optimizer == "hip_sparse":
        from diff_gaussian_rasterization import SparseGaussianAdam
        self.optimizer = SparseGaussianAdam(groups, lr=0.0, eps=1e-15)
    else:
        raise ValueError(optimizer)
    self.optimizer_kind = optimizer
    return self.optimizer


def add_densification_stats(self, viewspace_point_tensor, update_filter, radii=None):
    """reference gaussian_model.py:431-433 (+ train.py:159 max_radii2D when `radii` is given).  On the HIP device this is one
    kernel (csrc/densify.hip k_densify_stats; update_filter == radii > 0 as in the reference's call); the CPU branch
    (gloo tests) is the same arithmetic without boolean-mask indexing."""
    with torch.no_grad():
        g = viewspace_point_tensor.grad
        if g.is_cuda and radii is not None:
            from diff_gaussian_rasterization import _C
            P = int(g.shape[0])
            with _C.on_device(g.device):
                _C.check(_C.lib().gsr_densification_stats(P, _C.ptr(g.contiguous()), _C.ptr(radii.contiguous()),
                                                          _C.ptr(self.xyz_gradient_accum), _C.ptr(self.denom),
                                                          _C.ptr(self.max_radii2D), _C._stream()))
            return
        if radii is not None:
            torch.maximum(self.max_radii2D, radii.float(), out=self.max_radii2D)
        self.xyz_gradient_accum += torch.norm(g[:, :2], dim=-1, keepdim=True) * update_filter[:, None]
        self.denom += update_filter[:, None]


def _replace_params(self, new_tensors, new_moments=None):
    """Swap the six parameters (and their Adam moments) in the model and in the optimizer's groups / state."""
    new_params = []
    for gi, (attr, t) in enumerate(zip(_PARAM_ATTRS, new_tensors)):
        old = getattr(self, attr)
        p = nn.Parameter(t.requires_grad_(True))
        setattr(self, attr, p)
        new_params.append(p)
        opt = getattr(self, "optimizer", None)
        if opt is None:
            continue
        for group in opt.param_groups:
            if group["name"] == _GROUP_NAMES[gi]:
                st = opt.state.pop(old, None)
                group["params"][0] = p
                if st is not None:
                    if new_moments is not None and new_moments[gi] is not None:
                        st["exp_avg"], st["exp_avg_sq"] = new_moments[gi]
                    opt.state[p] = st
    return new_params


def reset_opacity(self):
    """reference gaussian_model.py:226-229: opacity <- min(opacity, 0.01), moments of the opacity group zeroed."""
    with torch.no_grad():
        new = _inverse_sigmoid(torch.min(self.get_opacity, torch.ones_like(self.get_opacity) * 0.01))
        old = self._opacity
        p = nn.Parameter(new.requires_grad_(True))
        self._opacity = p
        for group in self.optimizer.param_groups:
            if group["name"] == "opacity":
                st = self.optimizer.state.pop(old, None)
                group["params"][0] = p
                if st is not None:
                    st["exp_avg"] = torch.zeros_like(new)
                    st["exp_avg_sq"] = torch.zeros_like(new)
                    self.optimizer.state[p] = st


def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size, radii=None, seed=0, return_source=False):
    """reference gaussian_model.py:412-429.  Returns (n_keep, n_clone, n_split_sources[, source row of every new row])."""
    import ctypes as C
    from diff_gaussian_rasterization import _C
    lib = _C.lib()
    if not self._xyz.is_cuda:
        raise _C.GsrError("densify_and_prune runs in HIP kernels (no CPU path)")
    dev = self._xyz.device
    P = int(self._xyz.shape[0])
    with torch.no_grad(), torch.cuda.device(dev):
        params = [getattr(self, a).data.contiguous() for a in _PARAM_ATTRS]
        rows = [int(p.numel() // max(P, 1)) for p in params]
        moments = []
        for a in _PARAM_ATTRS:
            st = self.optimizer.state.get(getattr(self, a), None) if getattr(self, "optimizer", None) is not None else None
            moments.append((st["exp_avg"].contiguous(), st["exp_avg_sq"].contiguous()) if st and "exp_avg" in st else None)
        ws = torch.empty(lib.gsr_densify_workspace_bytes(P), dtype=torch.uint8, device=dev)
        counts = (C.c_int64 * 3)()
        _C.check(lib.gsr_densify_plan(P, _C.ptr(self.xyz_gradient_accum.contiguous()), _C.ptr(self.denom.contiguous()),
                                      _C.ptr(params[4]), _C.ptr(params[3]), float(max_grad), float(min_opacity),
                                      float(extent), float(self.percent_dense), 1 if max_screen_size else 0, _C.ptr(ws),
                                      ws.numel(), counts, _C._stream()))
        nk, nc, ns = int(counts[0]), int(counts[1]), int(counts[2])
        newP = nk + nc + 2 * ns
        outs = [torch.empty((newP,) + tuple(p.shape[1:]), dtype=torch.float32, device=dev) for p in params]
        out_m = [None if m is None else (torch.empty_like(o), torch.empty_like(o)) for m, o in zip(moments, outs)]
        ins18, outs18 = [], []
        for p, m, o, om in zip(params, moments, outs, out_m):
            ins18 += [p.data_ptr(), m[0].data_ptr() if m else None, m[1].data_ptr() if m else None]
            outs18 += [o.data_ptr(), om[0].data_ptr() if om else None, om[1].data_ptr() if om else None]
        vin = (C.c_void_p * 18)(*ins18)
        vout = (C.c_void_p * 18)(*outs18)
        rowf = (C.c_int32 * 6)(*rows)
        src = torch.empty(newP, dtype=torch.int32, device=dev) if return_source else None
        _C.check(lib.gsr_densify_apply(P, _C.ptr(ws), vin, vout, rowf, nk, nc, ns, int(seed) & 0xFFFFFFFF, _C.ptr(src),
                                       _C._stream()))
        _replace_params(self, outs, out_m)
        self.xyz_gradient_accum = torch.zeros((newP, 1), device=dev)      # densification_postfix :362-364
        self.denom = torch.zeros((newP, 1), device=dev)
        self.max_radii2D = torch.zeros((newP,), device=dev)
    return (nk, nc, ns, src) if return_source else (nk, nc, ns)


GaussianModel.training_setup = training_setup
GaussianModel.add_densification_stats = add_densification_stats
GaussianModel.reset_opacity = reset_opacity
GaussianModel.densify_and_prune = densify_and_prune
