"""`render()` with the contract of reference `gaussian_renderer/__init__.py:18-121`, calling the MI355X-native
rasterizer.  Same arguments, same branches (python-vs-native covariance / SH, `separate_sh`, `override_color`,
`use_trained_exp`), same returned keys; `"depth"` (the inverse-depth image the reference computes at :90,:101
and `train.py:126` reads) is returned as an additional key.
"""
import math

import torch

from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from scene_utils.sh import eval_sh


class PipelineParams:
    """Fields of reference arguments/__init__.py:65-71."""

    def __init__(self, convert_SHs_python=False, compute_cov3D_python=False, debug=False, antialiasing=False):
        self.convert_SHs_python = convert_SHs_python
        self.compute_cov3D_python = compute_cov3D_python
        self.debug = debug
        self.antialiasing = antialiasing


class RenderPackage(dict):
    """The dict `render()` returns.  `"visibility_filter"` (= `radii > 0`, reference :118-121) is computed when it is first
    looked at - by key, or by anything that enumerates the dict - instead of in every call: a training step whose
    densification statistics ride in the rasterizer's backward never reads it, and the launch is ~1 % of a 1080p step."""

    def _fill(self):
        if not dict.__contains__(self, "visibility_filter"):
            dict.__setitem__(self, "visibility_filter", dict.__getitem__(self, "radii") > 0)

    def __getitem__(self, key):
        if key == "visibility_filter":
            self._fill()
        return dict.__getitem__(self, key)


def _filled(name):
    def method(self, *a, **k):
        self._fill()
        return getattr(dict, name)(self, *a, **k)
    method.__name__ = name
    return method


for _name in ("get", "__contains__", "keys", "items", "values", "__iter__", "__len__", "__repr__", "copy", "__eq__", "pop"):
    setattr(RenderPackage, _name, _filled(_name))

_zeros = {}


def _screenspace_zeros(like):
    """A fresh leaf of zeros for every call (its .grad is what the callers read, reference :26-30) on top of ONE zero buffer
    per shape: the rasterizer never reads or writes its values, so the fill kernel of `torch.zeros_like` per call buys
    nothing."""
    key = (tuple(like.shape), like.dtype, like.device)
    z = _zeros.get(key)
    if z is None:
        _zeros.clear()
        z = _zeros[key] = torch.zeros_like(like, requires_grad=False)
    return z.detach().requires_grad_(True)


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, separate_sh=False,
           override_color=None, use_trained_exp=False, **rasterizer_kw):
    """`rasterizer_kw`: per-call extensions of this rasterizer, forwarded to `GaussianRasterizer.forward` (`fold`, `sh_ready_event`,
    `forward_mode`); none given = the reference's call forms, unchanged."""
    # zero tensor that receives the screen-space (NDC) gradient of the 2-D means (reference :26-30)
    # (a leaf here: its .grad is what the callers read; the reference's `+ 0` / retain_grad() pair gives the same .grad at
    # the price of one more launch per step)
    screenspace_points = _screenspace_zeros(pc.get_xyz)

    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=tanfovx,
        tanfovy=tanfovy,
        bg=bg_color,
        scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center,
        prefiltered=False,
        debug=pipe.debug,
        antialiasing=pipe.antialiasing,
    )
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)

    means3D = pc.get_xyz
    means2D = screenspace_points

    scales = rotations = cov3D_precomp = None
    # A model that hands out its RAW parameters (scene_utils.model.GaussianModel.get_raw_geometry) lets the rasterizer apply
    # exp / normalize / sigmoid inside its projection kernel (and chain them in its backward): no activation kernels at all
    # in the step.  Any other model object goes through the reference's getters.
    raw = getattr(pc, "get_raw_geometry", None)
    use_raw = raw is not None and not pipe.compute_cov3D_python
    if use_raw:
        scales, rotations, opacity = raw()
    else:
        opacity = pc.get_opacity
        if pipe.compute_cov3D_python:
            cov3D_precomp = pc.get_covariance(scaling_modifier)
        else:
            scales = pc.get_scaling
            rotations = pc.get_rotation

    shs = colors_precomp = dc = None
    if override_color is None:
        if pipe.convert_SHs_python:
            shs_view = pc.get_features.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
            dir_pp = pc.get_xyz - viewpoint_camera.camera_center.repeat(pc.get_features.shape[0], 1)
            dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
            sh2rgb = eval_sh(pc.active_sh_degree, shs_view, dir_pp_normalized)
            colors_precomp = torch.clamp_min(sh2rgb + 0.5, 0.0)
        elif separate_sh:
            dc, shs = pc.get_features_dc, pc.get_features_rest
        else:
            shs = pc.get_features
    else:
        colors_precomp = override_color

    if separate_sh:
        rendered_image, radii, depth_image = rasterizer(
            means3D=means3D, means2D=means2D, dc=dc, shs=shs, colors_precomp=colors_precomp, opacities=opacity,
            scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp, **({"raw_activations": True} if use_raw else {}),
            **rasterizer_kw)
    else:
        rendered_image, radii, depth_image = rasterizer(
            means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp, opacities=opacity,
            scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp, **({"raw_activations": True} if use_raw else {}),
            **rasterizer_kw)

    if use_trained_exp:
        exposure = pc.get_exposure_from_name(viewpoint_camera.image_name)
        rendered_image = torch.matmul(rendered_image.permute(1, 2, 0), exposure[:3, :3]).permute(2, 0, 1) + \
            exposure[:3, 3, None, None]

    return RenderPackage({"render": rendered_image,
                          "viewspace_points": screenspace_points,
                          "radii": radii,
                          "depth": depth_image})
