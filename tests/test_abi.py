"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol include/gsr.h declares; the
Python surface has the reference's names, field order and error behaviour (no compute without a GPU)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "gsr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gsr_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from diff_gaussian_rasterization import _C
    assert os.path.exists(_C.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_C.LIB_PATH)
    names = _declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/gsr.h but not exported"
    assert set(_C.EXPORTS) == set(names), set(_C.EXPORTS) ^ set(names)


def test_abi_version_and_size_queries():
    from diff_gaussian_rasterization import _C
    lib = _C.lib()
    assert lib.gsr_abi_version() == _C.ABI_VERSION == 7
    g1, g2 = lib.gsr_geometry_state_bytes(1000), lib.gsr_geometry_state_bytes(2000)
    assert 0 < g1 < g2 and g1 % 256 == 0
    assert lib.gsr_image_state_bytes(1920, 1080) >= 1920 * 1080 * 8
    b0 = lib.gsr_binning_state_bytes(1000, 256, 256, 0)
    b1 = lib.gsr_binning_state_bytes(1000, 256, 256, 100000)
    assert 0 < b0 < b1
    assert lib.gsr_backward_scratch_bytes(1000, 1000) >= 48000
    assert lib.gsr_geometry_state_bytes(0) > 0


def test_settings_fields_match_reference_order():
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    # reference gaussian_renderer/__init__.py:36-50
    assert GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "debug", "antialiasing")


def test_forward_signature_matches_reference_call_sites():
    import inspect
    from diff_gaussian_rasterization import GaussianRasterizer
    params = list(inspect.signature(GaussianRasterizer.forward).parameters)
    # keyword names used at reference gaussian_renderer/__init__.py:90-109
    for kw in ("means3D", "means2D", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp", "dc"):
        assert kw in params
    assert hasattr(GaussianRasterizer, "markVisible")


def _settings():
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    return GaussianRasterizationSettings(16, 16, 0.5, 0.5, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0,
                                         torch.zeros(3), False, False, False)


def test_argument_validation_raises_like_reference():
    from diff_gaussian_rasterization import GaussianRasterizer
    r = GaussianRasterizer(_settings())
    P = 4
    x, m2, op = torch.rand(P, 3), torch.zeros(P, 3), torch.rand(P, 1)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(x, m2, op, scales=torch.rand(P, 3), rotations=torch.rand(P, 4))
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(x, m2, op, shs=torch.rand(P, 1, 3), colors_precomp=torch.rand(P, 3), scales=torch.rand(P, 3),
          rotations=torch.rand(P, 4))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        r(x, m2, op, shs=torch.rand(P, 1, 3), scales=torch.rand(P, 3))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        r(x, m2, op, shs=torch.rand(P, 1, 3), scales=torch.rand(P, 3), rotations=torch.rand(P, 4),
          cov3D_precomp=torch.rand(P, 6))


def test_no_cpu_fallback():
    """The product path must fail loudly, not fall back, when asked to run without the HIP device."""
    from diff_gaussian_rasterization import GaussianRasterizer, _C
    r = GaussianRasterizer(_settings())
    P = 4
    with pytest.raises(_C.GsrError, match="no CPU path"):
        r(torch.rand(P, 3), torch.zeros(P, 3), torch.rand(P, 1), shs=torch.rand(P, 1, 3), scales=torch.rand(P, 3),
          rotations=torch.rand(P, 4))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "gaussian-splatting-slam_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(d, f)).read()
                assert "gs_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def test_low_level_names_of_the_published_extension_exist():
    from diff_gaussian_rasterization import _C
    for n in ("rasterize_gaussians", "rasterize_gaussians_backward", "mark_visible", "fusedssim", "fusedssim_backward"):
        assert callable(getattr(_C, n)), n
    import fused_ssim
    assert callable(fused_ssim.fused_ssim)
    from diff_gaussian_rasterization import SparseGaussianAdam   # reference train.py:37-41 probes this import
    assert SparseGaussianAdam is not None


def test_graft_entry_build_runs():
    """The driver's build check: __graft_entry__.build() compiles the HIP library for gfx950 (incremental make here) and loads
    it - kept under test so that an ABI bump cannot leave a stale version check behind."""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    g = importlib.import_module("__graft_entry__")
    g.build()
