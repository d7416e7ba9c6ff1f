"""GPU kernel arithmetic against vectors generated from the REFERENCE's own code (tests/golden/reference_helpers.npz, made by
tests/golden/make_reference_fixtures.py from reference utils/sh_utils.py:57-112 `eval_sh` and the clamp of
gaussian_renderer/__init__.py:74-79).  The rasterizer as a whole has no reference-held vector (its source is absent from the
reference tree), but the SH -> RGB half of the projection kernel does: the fixture's coefficients, positions and camera centre
go through every colour path of the library - `k_preprocess_fwd` with the rows staged through LDS (active degree = stored
degree) and unstaged (lower active degree), `shs` alone and the `dc` + `shs` call form, and the separate `k_shade` pass on the
caller's stream (gsr_forward_prepare_geometry + gsr_forward_shade) and on the library's side stream (gsr_forward_async with
GSR_SHADE_STREAM=1) - and the colours are read back out of the packed 48-B splat record.

Tolerance: fp32 evaluation of a 16-term sum with N(0, 1) coefficients against the float64 reference: |err| <= 1e-5 max(1, |rgb|)
(measured ~1e-6); clamp flags must equal `raw + 0.5 < 0` wherever the float64 value is not within 1e-5 of the threshold.
"""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch

from scene_utils import look_at_camera

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_helpers.npz"))
W = H = 128
FOV = 2.0          # ~115 degrees: six such views from one centre cover every direction


def _cams():
    eye = G["sh_campos"]
    out = []
    for d in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)):
        out.append(look_at_camera(eye, eye + np.array(d, dtype=np.float64), (0.0, 0.0, 1.0), FOV, W, H))
    return out


def _run(deg, form, path):
    """-> (rgb [P,3] with NaN where no view shaded the Gaussian, clamp flags [P,3] bool) collected over the six views."""
    from diff_gaussian_rasterization import _C, GaussianRasterizationSettings, _settings_struct, _gauss_struct, _stream
    from helpers import _view
    lib = _C.lib()
    dev = "cuda"
    P = G["sh_xyz"].shape[0]
    xyz = torch.tensor(G["sh_xyz"], dtype=torch.float32, device=dev).contiguous()
    sh = torch.tensor(G["sh_coeffs"], dtype=torch.float32, device=dev).contiguous()
    dc, rest = (sh[:, :1].contiguous(), sh[:, 1:].contiguous()) if form == "dc" else (None, sh)
    opac = torch.full((P, 1), 0.9, device=dev)
    scales = torch.full((P, 3), 0.03, device=dev)
    rots = torch.zeros(P, 4, device=dev)
    rots[:, 0] = 1.0
    rgb = torch.full((P, 3), float("nan"))
    clamp = torch.zeros(P, 3, dtype=torch.bool)
    old = os.environ.get("GSR_SHADE_STREAM")
    try:
        for cam in _cams():
            rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3, device=dev),
                                               1.0, cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), deg,
                                               torch.tensor(G["sh_campos"], dtype=torch.float32, device=dev), False, False, False)
            s, keep = _settings_struct(rs, dev)
            g = _gauss_struct(P, xyz, dc, rest, None, opac, scales, rots, None)
            geom = torch.zeros(lib.gsr_geometry_state_bytes(P), dtype=torch.uint8, device=dev)
            radii = torch.zeros(P, dtype=torch.int32, device=dev)
            if path == "fused":
                _C.check(lib.gsr_forward_prepare(C.byref(s), C.byref(g), _C.ptr(geom), geom.numel(), _C.ptr(radii), _stream()))
            elif path == "shade":
                _C.check(lib.gsr_forward_prepare_geometry(C.byref(s), C.byref(g), _C.ptr(geom), geom.numel(), _C.ptr(radii),
                                                          _stream()))
                _C.check(lib.gsr_forward_shade(C.byref(s), C.byref(g), _C.ptr(geom), _stream()))
            else:   # "aside": the colour pass on the library's side stream, inside the speculative forward
                os.environ["GSR_SHADE_STREAM"] = "1"
                cap = 1 << 16
                img = torch.zeros(lib.gsr_image_state_bytes(W, H), dtype=torch.uint8, device=dev)
                binning = torch.zeros(lib.gsr_binning_state_bytes(P, W, H, cap), dtype=torch.uint8, device=dev)
                color, invd = torch.empty(3, H, W, device=dev), torch.empty(1, H, W, device=dev)
                count = C.c_int64(-1)
                _C.check(lib.gsr_forward_async(C.byref(s), C.byref(g), _C.ptr(geom), geom.numel(), _C.ptr(radii),
                                               _C.ptr(binning), binning.numel(), cap, _C.ptr(img), img.numel(), _C.ptr(color),
                                               _C.ptr(invd), 0, 0, None, None, 1 if path == "aside_tlo" else 0, _stream(),
                                               C.byref(count)))
                assert 0 < count.value <= cap
            torch.cuda.synchronize()
            pv = [C.c_void_p() for _ in range(7)]
            lib.gsr_debug_geometry_views(_C.ptr(geom), P, *[C.byref(p) for p in pv])
            rec = _view(geom, pv[0].value, P * 12, torch.float32).view(P, 12)
            touched = _view(geom, pv[3].value, P, torch.int32) > 0
            cl = _view(geom, pv[6].value, P, torch.uint8)
            rgb[touched] = rec[touched][:, 7:10]
            for c in range(3):
                clamp[touched, c] = (cl[touched] >> c) & 1 > 0
    finally:
        if old is None:
            os.environ.pop("GSR_SHADE_STREAM", None)
        else:
            os.environ["GSR_SHADE_STREAM"] = old
    return rgb, clamp


@pytest.mark.parametrize("form", ["shs", "dc"])
@pytest.mark.parametrize("path", ["fused", "shade", "aside", "aside_tlo"])
@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_kernel_sh_to_rgb_matches_reference_eval_sh(deg, form, path):
    rgb, clamp = _run(deg, form, path)
    seen = ~torch.isnan(rgb[:, 0])
    assert int(seen.sum()) >= 250, int(seen.sum())          # (a point closer than the 0.2 near plane in every view is never shaded)
    exp = torch.tensor(G[f"sh_rgb_deg{deg}"])
    raw = torch.tensor(G[f"sh_raw_deg{deg}"])
    err = (rgb[seen].double() - exp[seen]).abs()
    tol = 1e-5 * torch.clamp(exp[seen].abs(), min=1.0)
    assert bool((err <= tol).all()), float((err / tol).max())
    decided = (raw[seen] + 0.5).abs() > 1e-5
    assert bool((clamp[seen] == (raw[seen] + 0.5 < 0))[decided].all())
    assert int((clamp[seen] & decided).sum()) > 0            # the fixture does exercise the clamp
