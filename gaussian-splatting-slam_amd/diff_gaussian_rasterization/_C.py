"""ctypes binding of libgsr_hip.so (C ABI in include/gsr.h).

This is the stub a maintainer of the reference would add in place of the pybind11 `_C` extension of the
absent `diff_gaussian_rasterization` submodule (reference gaussian_renderer/__init__.py:14).  There is no
CPU fallback: if the HIP library is missing every call raises.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must be imported first: the library shares torch's HIP runtime - same SONAME)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSR_LIB") or os.path.join(_HERE, "libgsr_hip.so")   # GSR_LIB: A/B a variant build

c_float_p = C.POINTER(C.c_float)


class gsr_settings(C.Structure):
    _fields_ = [
        ("image_height", C.c_int32), ("image_width", C.c_int32),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float),
        ("bg", C.c_void_p), ("scale_modifier", C.c_float),
        ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p),
        ("sh_degree", C.c_int32), ("campos", C.c_void_p),
        ("prefiltered", C.c_int32), ("debug", C.c_int32), ("antialiasing", C.c_int32),
    ]


class gsr_gaussians(C.Structure):
    _fields_ = [
        ("P", C.c_int32), ("sh_coeffs", C.c_int32),
        ("means3D", C.c_void_p), ("dc", C.c_void_p), ("shs", C.c_void_p), ("colors_precomp", C.c_void_p),
        ("opacities", C.c_void_p), ("scales", C.c_void_p), ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p),
        ("raw_activations", C.c_int32),
    ]


class gsr_grads(C.Structure):
    _fields_ = [
        ("dL_dmeans3D", C.c_void_p), ("dL_dmeans2D", C.c_void_p), ("dL_ddc", C.c_void_p), ("dL_dshs", C.c_void_p),
        ("dL_dcolors", C.c_void_p), ("dL_dopacities", C.c_void_p), ("dL_dscales", C.c_void_p),
        ("dL_drotations", C.c_void_p), ("dL_dcov3D", C.c_void_p),
        ("xyz_gradient_accum", C.c_void_p), ("denom", C.c_void_p), ("max_radii2D", C.c_void_p),
    ]


class gsr_fused_adam(C.Structure):
    _fields_ = [
        ("exp_avg", C.c_void_p * 6), ("exp_avg_sq", C.c_void_p * 6), ("lr", C.c_float * 6), ("step", C.c_int64 * 6),
        ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double), ("sparse", C.c_int32),
        ("dynamic", C.c_void_p),
    ]


EXPORTS = {
    # name: (restype, argtypes)
    "gsr_abi_version": (C.c_int, []),
    "gsr_last_error": (C.c_char_p, []),
    "gsr_geometry_state_bytes": (C.c_size_t, [C.c_int32]),
    "gsr_image_state_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "gsr_binning_state_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int64]),
    "gsr_backward_scratch_bytes": (C.c_size_t, [C.c_int32, C.c_int64]),
    "gsr_forward_prepare": (C.c_int64, [C.POINTER(gsr_settings), C.POINTER(gsr_gaussians), C.c_void_p, C.c_size_t,
                                        C.c_void_p, C.c_void_p]),
    "gsr_forward_prepare_geometry": (C.c_int64, [C.POINTER(gsr_settings), C.POINTER(gsr_gaussians), C.c_void_p,
                                                 C.c_size_t, C.c_void_p, C.c_void_p]),
    "gsr_forward_shade": (C.c_int, [C.POINTER(gsr_settings), C.POINTER(gsr_gaussians), C.c_void_p, C.c_void_p]),
    "gsr_forward_render": (C.c_int, [C.POINTER(gsr_settings), C.POINTER(gsr_gaussians), C.c_void_p, C.c_void_p,
                                     C.c_size_t, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                     C.c_int32, C.c_void_p]),
    "gsr_forward_render_shade": (C.c_int, [C.POINTER(gsr_settings), C.POINTER(gsr_gaussians), C.c_void_p, C.c_void_p,
                                           C.c_size_t, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                           C.c_int32, C.c_void_p, C.c_void_p]),
    "gsr_forward_async": (C.c_int, [C.POINTER(gsr_settings), C.POINTER(gsr_gaussians), C.c_void_p, C.c_size_t, C.c_void_p,
                                    C.c_void_p, C.c_size_t, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                    C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                    C.POINTER(C.c_int64)]),
    "gsr_forward_async_culled": (C.c_int, [C.POINTER(gsr_settings), C.POINTER(gsr_gaussians), C.c_void_p, C.c_size_t, C.c_void_p,
                                    C.c_void_p, C.c_size_t, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                    C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                    C.POINTER(C.c_int64), C.c_void_p, C.c_int32]),
    "gsr_forward_rerender": (C.c_int, [C.POINTER(gsr_settings), C.POINTER(gsr_gaussians), C.c_void_p, C.c_void_p,
                                       C.c_size_t, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int32,
                                       C.c_int32, C.c_void_p, C.c_void_p]),
    "gsr_backward": (C.c_int, [C.POINTER(gsr_settings), C.POINTER(gsr_gaussians), C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                               C.POINTER(gsr_grads), C.c_void_p]),
    "gsr_backward_adam": (C.c_int, [C.POINTER(gsr_settings), C.POINTER(gsr_gaussians), C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                    C.POINTER(gsr_grads), C.POINTER(gsr_fused_adam), C.c_void_p]),
    "gsr_adam_step_culled_rows": (C.c_int, [C.POINTER(gsr_gaussians), C.c_void_p, C.c_int64, C.POINTER(gsr_fused_adam),
                                            C.c_void_p]),
    "gsr_mark_visible": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gsr_debug_geometry_views": (C.c_int, [C.c_void_p, C.c_int32] + [C.POINTER(C.c_void_p)] * 7),
    "gsr_debug_wave_reduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "gsr_debug_wave_reduce_pk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "gsr_debug_mx_reduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "gsr_debug_binning_views": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_void_p)]),
    "gsr_debug_count_pairs": (C.c_int, [C.POINTER(gsr_settings), C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                        C.c_void_p]),
    "gsr_debug_radix_tmp_bytes": (C.c_size_t, [C.c_int64]),
    "gsr_debug_radix_sort": (C.c_int, [C.c_void_p] * 6 + [C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gsr_debug_image_views": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p),
                                        C.POINTER(C.c_void_p)]),
    "gsr_debug_set_flags_min_r": (C.c_int64, [C.c_int64]),
    "gsr_debug_walk_views": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_void_p)]),
    "gsr_fused_ssim_forward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float] + [C.c_void_p] * 7),
    "gsr_fused_ssim_backward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 8),
    "gsr_fused_loss_blocks": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "gsr_adam_set_dynamic": (C.c_int, [C.POINTER(gsr_fused_adam), C.c_void_p, C.c_void_p]),
    "gsr_l1_mean_blocks": (C.c_int32, []),
    "gsr_l1_mean_forward": (C.c_int, [C.c_int64, C.c_float] + [C.c_void_p] * 6),
    "gsr_l1_mean_backward": (C.c_int, [C.c_int64, C.c_float] + [C.c_void_p] * 6),
    "gsr_fused_l1_ssim_forward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float] +
                                  [C.c_void_p] * 8),
    "gsr_fused_l1_ssim_backward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_float] + [C.c_void_p] * 8),
    "gsr_adam_step": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p]),
    "gsr_sparse_adam_step": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int64, C.c_void_p, C.c_double, C.c_double, C.c_double,
                                       C.c_void_p]),
    "gsr_densification_stats": (C.c_int, [C.c_int64] + [C.c_void_p] * 6),
    "gsr_gaussian_activations_forward": (C.c_int, [C.c_int32] + [C.c_void_p] * 7),
    "gsr_gaussian_activations_backward": (C.c_int, [C.c_int32] + [C.c_void_p] * 10),
    "gsr_densify_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "gsr_densify_plan": (C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                   C.c_float, C.c_float, C.c_int32, C.c_void_p, C.c_size_t, C.POINTER(C.c_int64),
                                   C.c_void_p]),
    "gsr_densify_apply": (C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_int64,
                                    C.c_int64, C.c_int64, C.c_uint32, C.c_void_p, C.c_void_p]),
    "gsr_sh_rank1_expand": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    "gsr_sh_rank1_adam": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                    C.c_void_p, C.POINTER(gsr_fused_adam), C.c_void_p]),
    "gsr_profile_enable": (None, [C.c_int32]),
    "gsr_profile_reset": (None, []),
    "gsr_profile_read": (C.c_int32, [C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int32]),
}

ADAM_DYNAMIC_FLOATS = 18     # GSR_ADAM_DYNAMIC_FLOATS
ABI_VERSION = 7      # GSR_ABI_VERSION of include/gsr.h
_lib = None


class GsrError(RuntimeError):
    pass


def lib():
    """Loads libgsr_hip.so once.  Raises (never falls back) when the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GsrError(
                f"{LIB_PATH} not found: the HIP rasterizer is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C gaussian-splatting-slam_amd/csrc`). "
                "There is no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in EXPORTS.items():
            f = getattr(l, name)         # AttributeError if a declared symbol is not exported
            f.restype = res
            f.argtypes = args
        if l.gsr_abi_version() != ABI_VERSION:
            raise GsrError(f"libgsr_hip.so ABI {l.gsr_abi_version()} != {ABI_VERSION}: rebuild (make -C csrc)")
        _lib = l
    return _lib


def last_error() -> str:
    return lib().gsr_last_error().decode("utf-8", "replace")


def check(rc: int):
    if rc < 0:
        raise GsrError(f"libgsr_hip error {rc}: {last_error()}")
    return rc


def ptr(t):
    """device pointer of a tensor, or None"""
    return None if t is None else C.c_void_p(t.data_ptr())


def profile_read():
    l = lib()
    n = l.gsr_profile_read(None, None, None, 0)
    names = (C.c_char_p * n)()
    ms = (C.c_double * n)()
    calls = (C.c_int64 * n)()
    n = l.gsr_profile_read(names, ms, calls, n)
    return {names[i].decode(): (ms[i], calls[i]) for i in range(n)}


def raw_stream(device_index=None):
    """hipStream_t of torch's current stream as an int (no Stream object is built: this sits on every call's path)."""
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device() if device_index is None else device_index)


def _stream():
    return C.c_void_p(raw_stream())


class on_device:
    """`with torch.cuda.device(dev)` that does nothing at all when `dev` already is the current device (the usual case: one
    process per GPU) - the stock context manager costs ~10 us per use, several uses per training step."""
    __slots__ = ("idx", "prev")

    def __init__(self, dev):
        self.idx = dev.index if isinstance(dev, torch.device) else dev
        self.prev = None

    def __enter__(self):
        if self.idx is not None:
            cur = torch.cuda.current_device()
            if cur != self.idx:
                self.prev = cur
                torch.cuda.set_device(self.idx)
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)
        return False


# ---- the two functions reference utils/loss_utils.py:16-19 imports from `diff_gaussian_rasterization._C` ----
def _ssim_forward(C1, C2, img1, img2, want_partials):
    if not img1.is_cuda:
        raise GsrError("fusedssim needs tensors on the HIP device (no CPU path)")
    img1 = img1.detach().float().contiguous()
    img2 = img2.detach().float().contiguous()
    H, W = int(img1.shape[-2]), int(img1.shape[-1])
    planes = img1.numel() // (H * W) if H * W else 0
    out = torch.empty_like(img1)
    parts = [torch.empty_like(img1) for _ in range(3)] if want_partials else [None, None, None]
    with torch.cuda.device(img1.device):
        check(lib().gsr_fused_ssim_forward(planes, H, W, float(C1), float(C2), ptr(img1), ptr(img2), ptr(out),
                                           ptr(parts[0]), ptr(parts[1]), ptr(parts[2]), _stream()))
    return out, parts, img1, img2


def fusedssim(C1, C2, img1, img2):
    """-> ssim_map, same shape as img1 (reference utils/loss_utils.py:27)."""
    return _ssim_forward(C1, C2, img1, img2, False)[0]


def fusedssim_backward(C1, C2, img1, img2, dL_dmap, partials=None):
    """-> dL/dimg1 (reference utils/loss_utils.py:37).  `partials` (from a forward that kept them) avoids recomputation."""
    if partials is None:
        _, partials, img1, img2 = _ssim_forward(C1, C2, img1, img2, True)
    else:
        img1 = img1.detach().float().contiguous()
        img2 = img2.detach().float().contiguous()
    H, W = int(img1.shape[-2]), int(img1.shape[-1])
    planes = img1.numel() // (H * W) if H * W else 0
    g = dL_dmap.detach().float().expand_as(img1).contiguous()
    out = torch.empty_like(img1)
    with torch.cuda.device(img1.device):
        check(lib().gsr_fused_ssim_backward(planes, H, W, ptr(img1), ptr(img2), ptr(g), ptr(partials[0]),
                                            ptr(partials[1]), ptr(partials[2]), ptr(out), _stream()))
    return out


# ---- the low-level call forms of the absent module's `_C` (SURVEY.md 8b "Native surface"; nothing in the reference calls them
# directly, they are kept so code written against the published extension keeps working) ----
def rasterize_gaussians(bg, means3D, colors_precomp, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                        projmatrix, tanfovx, tanfovy, image_height, image_width, sh, degree, campos, prefiltered,
                        antialiasing, debug):
    """-> (num_rendered, color[3,H,W], radii[P], geomBuffer, binningBuffer, imgBuffer, invdepth[1,H,W]); empty tensors stand
    for absent inputs, as in the published extension."""
    import diff_gaussian_rasterization as dgr

    def opt(t):
        return None if (t is None or t.numel() == 0) else t.detach().float().contiguous()
    rs = dgr.GaussianRasterizationSettings(int(image_height), int(image_width), float(tanfovx), float(tanfovy), bg,
                                           float(scale_modifier), viewmatrix, projmatrix, int(degree), campos,
                                           bool(prefiltered), bool(debug), bool(antialiasing))
    dev = means3D.device
    P, H, W = int(means3D.shape[0]), int(image_height), int(image_width)
    m3, col, op, sc, ro, cov, shs = opt(means3D), opt(colors_precomp), opt(opacity), opt(scales), opt(rotations), \
        opt(cov3D_precomp), opt(sh)
    l = lib()
    with torch.cuda.device(dev):
        s, keep = dgr._settings_struct(rs, dev)
        g = dgr._gauss_struct(P, m3, None, shs, col, op, sc, ro, cov)
        color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
        invdepth = torch.empty(1, H, W, dtype=torch.float32, device=dev)
        radii = torch.zeros(P, dtype=torch.int32, device=dev)
        geom = torch.empty(l.gsr_geometry_state_bytes(P), dtype=torch.uint8, device=dev)
        img = torch.empty(l.gsr_image_state_bytes(W, H), dtype=torch.uint8, device=dev)
        R = check(l.gsr_forward_prepare(C.byref(s), C.byref(g), ptr(geom), geom.numel(), ptr(radii), _stream()))
        binning = torch.empty(l.gsr_binning_state_bytes(P, W, H, R), dtype=torch.uint8, device=dev)
        check(l.gsr_forward_render(C.byref(s), C.byref(g), ptr(geom), ptr(binning), binning.numel(), R, ptr(img),
                                   img.numel(), ptr(color), ptr(invdepth), 1, _stream()))
    return R, color, radii, geom, binning, img, invdepth


def rasterize_gaussians_backward(bg, means3D, radii, colors_precomp, opacities, scales, rotations, scale_modifier,
                                 cov3D_precomp, viewmatrix, projmatrix, tanfovx, tanfovy, dL_dcolor, dL_dinvdepth, sh, degree,
                                 campos, geomBuffer, num_rendered, binningBuffer, imgBuffer, antialiasing, debug):
    """-> (dL_dmeans2D[P,3], dL_dcolors[P,3], dL_dopacity[P,1], dL_dmeans3D[P,3], dL_dcov3D[P,6], dL_dsh[P,M,3],
    dL_dscales[P,3], dL_drotations[P,4]) - the published order; absent inputs give empty gradient tensors."""
    import diff_gaussian_rasterization as dgr

    def opt(t):
        return None if (t is None or t.numel() == 0) else t.detach().float().contiguous()
    dev = means3D.device
    P = int(means3D.shape[0])
    H, W = int(dL_dcolor.shape[-2]), int(dL_dcolor.shape[-1])
    rs = dgr.GaussianRasterizationSettings(H, W, float(tanfovx), float(tanfovy), bg, float(scale_modifier), viewmatrix,
                                           projmatrix, int(degree), campos, False, bool(debug), bool(antialiasing))
    m3, col, op, sc, ro, cov, shs = opt(means3D), opt(colors_precomp), opt(opacities), opt(scales), opt(rotations), \
        opt(cov3D_precomp), opt(sh)
    l = lib()

    def like(t, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=dev) if t is not None else None
    with torch.cuda.device(dev):
        s, keep = dgr._settings_struct(rs, dev)
        g = dgr._gauss_struct(P, m3, None, shs, col, op, sc, ro, cov)
        d_m3 = torch.empty(P, 3, dtype=torch.float32, device=dev)
        d_m2 = torch.empty(P, 3, dtype=torch.float32, device=dev)
        d_op = torch.empty(P, 1, dtype=torch.float32, device=dev)
        d_sh, d_col = like(shs, *(shs.shape if shs is not None else ())), like(col, P, 3)
        d_sc, d_ro, d_cov = like(sc, P, 3), like(ro, P, 4), like(cov, P, 6)
        scratch = torch.empty(l.gsr_backward_scratch_bytes(P, int(num_rendered)), dtype=torch.uint8, device=dev)
        gr = gsr_grads(*[None if t is None else t.data_ptr() for t in (d_m3, d_m2, None, d_sh, d_col, d_op, d_sc, d_ro, d_cov)])
        check(l.gsr_backward(C.byref(s), C.byref(g), ptr(radii), ptr(geomBuffer), ptr(binningBuffer), ptr(imgBuffer),
                             int(num_rendered), ptr(opt(dL_dcolor)), ptr(opt(dL_dinvdepth)), ptr(scratch), scratch.numel(),
                             C.byref(gr), _stream()))
    e = torch.empty(0, device=dev)
    return (d_m2, d_col if d_col is not None else e, d_op, d_m3, d_cov if d_cov is not None else e,
            d_sh if d_sh is not None else e, d_sc if d_sc is not None else e, d_ro if d_ro is not None else e)


def mark_visible(positions, viewmatrix, projmatrix=None):
    """bool[P] (published `_C.mark_visible(positions, viewmatrix, projmatrix)`; the projection matrix is not needed for the
    near-plane test)."""
    positions = positions.detach().float().contiguous()
    P = int(positions.shape[0])
    present = torch.zeros(P, dtype=torch.uint8, device=positions.device)
    if P:
        vm = viewmatrix.detach().float().contiguous().to(positions.device)
        with torch.cuda.device(positions.device):
            check(lib().gsr_mark_visible(P, ptr(positions), ptr(vm), ptr(present), _stream()))
    return present.bool()
