"""MI355X-native drop-in for the `diff_gaussian_rasterization` module the reference imports at
`gaussian_renderer/__init__.py:14`:

    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

Same Python surface: `GaussianRasterizationSettings` (13 fields, order of reference
`gaussian_renderer/__init__.py:36-50`), `GaussianRasterizer(raster_settings)` whose call returns
`(color[3,H,W], radii[P] int32, invdepth[1,H,W])` (reference :90-109), `markVisible`, the `means2D.grad`
side channel (reference :26-30; consumed at `scene/gaussian_model.py:431-433`).  All arithmetic runs in
hand-written HIP kernels behind the C ABI of `include/gsr.h`; there is no PyTorch/CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    antialiasing: bool = False


# One-shot hand-off used by the view-sharded data-parallel trainer (scene_utils/trainer.py): when set, the NEXT forward runs
# the geometry stages first, makes the current stream wait for this event (the SH coefficients' all-reduce + Adam update,
# in flight on another stream), and only then evaluates the colours (gsr_forward_prepare_geometry / gsr_forward_shade).
_sh_ready_event = None


def defer_sh_until(event):
    """`event`: a recorded torch.cuda.Event after which `dc` / `shs` hold this step's values (None cancels)."""
    global _sh_ready_event
    _sh_ready_event = event


# statistics of the most recent forward call (bench.py reports the measured num_rendered with every number)
last_call_stats = {"num_rendered": 0}


def _dump(path, rs, *tensors):
    """debug=True failure snapshot (reference README.md:168-169 `snapshot_fw.dump` / `snapshot_bw.dump`)."""
    try:
        torch.save({"settings": {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in rs._asdict().items()},
                    "tensors": [None if t is None else t.detach().cpu() for t in tensors]}, path)
        print(f"\nAn error occured in the rasterizer. Writing {path} for debugging.")
    except Exception as e:  # never mask the original error
        print(f"could not write {path}: {e}")


def _f32c(t):
    if t is None or t.numel() == 0:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _settings_struct(rs: GaussianRasterizationSettings, device):
    bg = _f32c(rs.bg.to(device))
    vm = _f32c(rs.viewmatrix.to(device))
    pm = _f32c(rs.projmatrix.to(device))
    cp = _f32c(rs.campos.to(device))
    s = _C.gsr_settings(
        int(rs.image_height), int(rs.image_width), float(rs.tanfovx), float(rs.tanfovy),
        bg.data_ptr(), float(rs.scale_modifier), vm.data_ptr(), pm.data_ptr(), int(rs.sh_degree), cp.data_ptr(),
        int(bool(rs.prefiltered)), int(bool(rs.debug)), int(bool(rs.antialiasing)))
    return s, (bg, vm, pm, cp)   # keep the tensors alive


def _gauss_struct(P, means3D, dc, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, raw_activations=False):
    def p(t):
        return None if t is None else t.data_ptr()
    return _C.gsr_gaussians(
        int(P), int(sh.shape[1]) if sh is not None else 0,
        p(means3D), p(dc), p(sh), p(colors_precomp), p(opacities), p(scales), p(rotations), p(cov3D_precomp),
        1 if raw_activations else 0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _RasterizeGaussians(torch.autograd.Function):
    """Input order = gradient order of the reference's autograd Function, with `dc` inserted
    before `sh` for the `separate_sh` call form (reference gaussian_renderer/__init__.py:90-99)."""

    @staticmethod
    def forward(ctx, means3D, means2D, dc, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                raster_settings, raw_activations=False, for_backward=True):
        lib = _C.lib()
        raw_activations = bool(raw_activations) and cov3D_precomp is None
        if not means3D.is_cuda:
            raise _C.GsrError("GaussianRasterizer needs tensors on the HIP device (no CPU path)")
        dev = means3D.device
        rs = raster_settings
        P = int(means3D.shape[0])
        H, W = int(rs.image_height), int(rs.image_width)
        means3D = _f32c(means3D) if P > 0 else means3D
        dc, sh, colors_precomp = _f32c(dc), _f32c(sh), _f32c(colors_precomp)
        opacities, scales, rotations, cov3D_precomp = _f32c(opacities), _f32c(scales), _f32c(rotations), \
            _f32c(cov3D_precomp)
        # (inside forward() grad mode is always off and needs_input_grad ignores an outer torch.no_grad(): whether a backward
        # can follow is decided by the caller, rasterize_gaussians(), and arrives as `for_backward`)
        needs_grad = bool(for_backward)

        with torch.cuda.device(dev):
            color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
            invdepth = torch.empty(1, H, W, dtype=torch.float32, device=dev)
            radii = torch.empty(P, dtype=torch.int32, device=dev)     # every entry is written by the projection kernel
            s, keep = _settings_struct(rs, dev)
            g = _gauss_struct(P, means3D, dc, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                              raw_activations)
            geom = torch.empty(lib.gsr_geometry_state_bytes(P), dtype=torch.uint8, device=dev)
            img = torch.empty(lib.gsr_image_state_bytes(W, H), dtype=torch.uint8, device=dev)
            global _sh_ready_event
            ev, _sh_ready_event = _sh_ready_event, None
            try:
                if ev is not None and colors_precomp is None:
                    # split forward: geometry stages, emission + tile sort, THEN wait for the SH update, shade, composite
                    R = _C.check(lib.gsr_forward_prepare_geometry(C.byref(s), C.byref(g), _C.ptr(geom), geom.numel(),
                                                                  _C.ptr(radii), _stream()))
                    binning = torch.empty(lib.gsr_binning_state_bytes(P, W, H, R), dtype=torch.uint8, device=dev)
                    evh = C.c_void_p(ev.cuda_event)
                    _C.check(lib.gsr_forward_render_shade(C.byref(s), C.byref(g), _C.ptr(geom), _C.ptr(binning),
                                                          binning.numel(), R, _C.ptr(img), img.numel(), _C.ptr(color),
                                                          _C.ptr(invdepth), 1 if needs_grad else 0, evh, _stream()))
                else:
                    if ev is not None:
                        torch.cuda.current_stream().wait_event(ev)
                    R = _C.check(lib.gsr_forward_prepare(C.byref(s), C.byref(g), _C.ptr(geom), geom.numel(),
                                                         _C.ptr(radii), _stream()))
                    binning = torch.empty(lib.gsr_binning_state_bytes(P, W, H, R), dtype=torch.uint8, device=dev)
                    _C.check(lib.gsr_forward_render(C.byref(s), C.byref(g), _C.ptr(geom), _C.ptr(binning), binning.numel(),
                                                    R, _C.ptr(img), img.numel(), _C.ptr(color), _C.ptr(invdepth),
                                                    1 if needs_grad else 0, _stream()))
            except _C.GsrError:
                if rs.debug:   # reference README.md:168-169: with --debug a failing rasterizer call dumps its inputs
                    _dump("snapshot_fw.dump", rs, means3D, dc, sh, colors_precomp, opacities, scales, rotations,
                          cov3D_precomp)
                raise
        last_call_stats["num_rendered"] = int(R)
        ctx.raster_settings = rs
        ctx.raw_activations = raw_activations
        ctx.num_rendered = R
        ctx.has = (dc is not None, sh is not None, colors_precomp is not None, scales is not None,
                   cov3D_precomp is not None)
        ctx.save_for_backward(means3D, dc, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, radii,
                              geom, binning, img)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)     # an unused inverse-depth output reaches backward as None, not as zeros
        return color, radii, invdepth

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_invdepth):
        lib = _C.lib()
        (means3D, dc, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, radii, geom, binning,
         img) = ctx.saved_tensors
        rs = ctx.raster_settings
        R = ctx.num_rendered
        dev = means3D.device
        P = int(means3D.shape[0])
        H, W = int(rs.image_height), int(rs.image_width)
        grad_color = _f32c(grad_color) if grad_color is not None else torch.zeros(3, H, W, device=dev)
        if grad_color is None:
            grad_color = torch.zeros(3, H, W, device=dev)
        grad_invdepth = _f32c(grad_invdepth)

        def like(t, *shape):
            return torch.empty(*shape, dtype=torch.float32, device=dev) if t is not None else None

        with torch.cuda.device(dev):
            d_means3D = torch.empty(P, 3, dtype=torch.float32, device=dev)
            d_means2D = torch.empty(P, 3, dtype=torch.float32, device=dev)
            d_opac = torch.empty(opacities.shape if opacities is not None else (P, 1), dtype=torch.float32, device=dev)
            d_dc = like(dc, *(dc.shape if dc is not None else ()))
            d_sh = like(sh, *(sh.shape if sh is not None else ()))
            d_col = like(colors_precomp, P, 3)
            d_scales = like(scales, P, 3)
            d_rot = like(rotations, P, 4)
            d_cov = like(cov3D_precomp, P, 6)
            if P > 0:
                s, keep = _settings_struct(rs, dev)
                g = _gauss_struct(P, means3D, dc, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                  ctx.raw_activations)
                scratch = torch.empty(lib.gsr_backward_scratch_bytes(P, R), dtype=torch.uint8, device=dev)
                gr = _C.gsr_grads(*[None if t is None else t.data_ptr() for t in
                                    (d_means3D, d_means2D, d_dc, d_sh, d_col, d_opac, d_scales, d_rot, d_cov)])
                try:
                    _C.check(lib.gsr_backward(C.byref(s), C.byref(g), _C.ptr(radii), _C.ptr(geom), _C.ptr(binning),
                                              _C.ptr(img), R, _C.ptr(grad_color), _C.ptr(grad_invdepth),
                                              _C.ptr(scratch), scratch.numel(), C.byref(gr), _stream()))
                except _C.GsrError:
                    if rs.debug:
                        _dump("snapshot_bw.dump", rs, means3D, dc, sh, colors_precomp, opacities, scales, rotations,
                              cov3D_precomp, grad_color, grad_invdepth, radii)
                    raise
        return (d_means3D, d_means2D, d_dc, d_sh, d_col, d_opac, d_scales, d_rot, d_cov, None, None, None)


def rasterize_gaussians(means3D, means2D, dc, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, raw_activations=False):
    # forward-only render (torch.no_grad(), reference render.py:49, or no input that requires grad): the library then skips
    # what only a backward would need
    tensors = (means3D, means2D, dc, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp)
    for_backward = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)
    return _RasterizeGaussians.apply(means3D, means2D, dc, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, raw_activations, for_backward)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """bool[P]: Gaussians in front of the near plane of this rasterizer's camera."""
        lib = _C.lib()
        with torch.no_grad():
            positions = _f32c(positions)
            P = int(positions.shape[0])
            present = torch.zeros(P, dtype=torch.uint8, device=positions.device)
            if P > 0:
                vm = _f32c(self.raster_settings.viewmatrix.to(positions.device))
                with torch.cuda.device(positions.device):
                    _C.check(lib.gsr_mark_visible(P, _C.ptr(positions), _C.ptr(vm), _C.ptr(present), _stream()))
            return present.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, dc=None, raw_activations=False):
        """Arguments of the reference's call (gaussian_renderer/__init__.py:90-109).  `raw_activations=True` (an extension,
        keyword only in spirit): `opacities`, `scales`, `rotations` are the model's RAW parameters; sigmoid / exp / normalize
        are applied inside the projection kernel and the returned gradients are w.r.t. the raw parameters."""
        def none_if_empty(t):
            return None if (t is None or t.numel() == 0) else t
        shs, colors_precomp, dc = none_if_empty(shs), none_if_empty(colors_precomp), none_if_empty(dc)
        scales, rotations, cov3D_precomp = none_if_empty(scales), none_if_empty(rotations), none_if_empty(cov3D_precomp)
        P = int(means3D.shape[0])
        if P > 0:
            if ((shs is None and dc is None) and colors_precomp is None) or \
                    ((shs is not None or dc is not None) and colors_precomp is not None):
                raise Exception('Please provide excatly one of either SHs or precomputed colors!')
            if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                    ((scales is not None or rotations is not None) and cov3D_precomp is not None):
                raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        return rasterize_gaussians(means3D, means2D, dc, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, self.raster_settings, raw_activations)


from .sparse_adam import SparseGaussianAdam, FusedAdam  # noqa: E402,F401   (reference train.py:37-41)
