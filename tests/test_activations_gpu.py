"""Fused parameter activations (gsr_gaussian_activations_*) against the three PyTorch ops the reference's model uses
(scene/gaussian_model.py:38-46: torch.exp, torch.nn.functional.normalize, torch.sigmoid), forward and backward.
Tolerance: fp32 round-off of a handful of operations (rtol 2e-6 / atol 1e-7 forward, 1e-5 relative on gradients)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _raw(P, seed):
    g = torch.Generator().manual_seed(seed)
    s = (torch.randn(P, 3, generator=g) * 1.5 - 3.0)
    q = torch.randn(P, 4, generator=g) * torch.rand(P, 1, generator=g) * 3.0
    o = torch.randn(P, 1, generator=g) * 3.0
    return s, q, o


def test_fused_activations_match_torch_ops():
    from scene_utils.activations import gaussian_activations
    for P in (1, 255, 4099):
        s0, q0, o0 = _raw(P, 3 + P)
        a = [t.clone().cuda().requires_grad_(True) for t in (s0, q0, o0)]
        b = [t.clone().cuda().requires_grad_(True) for t in (s0, q0, o0)]
        fs, fq, fo = gaussian_activations(*a)
        ts, tq, to = torch.exp(b[0]), torch.nn.functional.normalize(b[1]), torch.sigmoid(b[2])
        for f, t in ((fs, ts), (fq, tq), (fo, to)):
            assert f.shape == t.shape
            torch.testing.assert_close(f, t, rtol=2e-6, atol=1e-7)
        g = torch.Generator().manual_seed(11)
        ws, wq, wo = (torch.randn(t.shape, generator=g).cuda() for t in (ts, tq, to))
        ((fs * ws).sum() + (fq * wq).sum() + (fo * wo).sum()).backward()
        ((ts * ws).sum() + (tq * wq).sum() + (to * wo).sum()).backward()
        for x, y in zip(a, b):
            torch.testing.assert_close(x.grad, y.grad, rtol=1e-5, atol=1e-6)


def test_fused_activations_partial_gradients_and_degenerate_quaternion():
    from scene_utils.activations import gaussian_activations
    s0, q0, o0 = _raw(300, 5)
    q0[7] = 0.0                                   # |q| < eps: forward 0/eps = 0, gradient g/eps like F.normalize
    a = [t.clone().cuda().requires_grad_(True) for t in (s0, q0, o0)]
    b = [t.clone().cuda().requires_grad_(True) for t in (s0, q0, o0)]
    fs, fq, fo = gaussian_activations(*a)
    assert torch.equal(fq[7], torch.zeros(4, device="cuda"))
    (fq[:, 1] * 2.0).sum().backward()             # only the rotation output is used: the other two gradients arrive as None
    (torch.nn.functional.normalize(b[1])[:, 1] * 2.0).sum().backward()
    keep = torch.ones(300, dtype=torch.bool); keep[7] = False
    torch.testing.assert_close(a[1].grad[keep.cuda()], b[1].grad[keep.cuda()], rtol=1e-5, atol=1e-6)
    assert torch.isfinite(a[1].grad).all()
    assert torch.count_nonzero(a[0].grad) == 0 and torch.count_nonzero(a[2].grad) == 0


def test_render_raw_parameter_path_matches_getter_path():
    """render() hands the RAW parameters of a model that offers get_raw_geometry() to the rasterizer
    (`raw_activations=True`: exp / normalize / sigmoid inside the projection kernel, chained in its backward); image and
    parameter gradients equal those of the reference's getter path (torch.exp / F.normalize / torch.sigmoid + autograd)."""
    from gaussian_renderer import PipelineParams, render
    from scene_utils.cameras import look_at_camera
    from scene_utils.model import GaussianModel
    from scene_utils.synthetic import make_gaussians

    raw = make_gaussians(1500, 3, seed=9)
    cam = look_at_camera((0.0, -4.0, 0.5), (0, 0, 0), (0, 0, 1), 1.0, 96, 64).to("cuda")
    bg = torch.tensor([0.1, 0.2, 0.3], device="cuda")
    pipe = PipelineParams()

    class GetterOnly:                                  # the reference's interface: no get_activated
        def __init__(self, m):
            self._m = m
        active_sh_degree = property(lambda self: self._m.active_sh_degree)
        max_sh_degree = property(lambda self: self._m.max_sh_degree)
        get_xyz = property(lambda self: self._m.get_xyz)
        get_opacity = property(lambda self: self._m.get_opacity)
        get_scaling = property(lambda self: self._m.get_scaling)
        get_rotation = property(lambda self: self._m.get_rotation)
        get_features = property(lambda self: self._m.get_features)

    outs = []
    for wrap in (lambda m: m, GetterOnly):
        m = GaussianModel.from_raw(raw.to("cuda"))
        pkg = render(cam, wrap(m), pipe, bg)
        pkg["render"].square().sum().backward()
        outs.append((pkg["render"].detach(), [p.grad.clone() for p in (m._xyz, m._scaling, m._rotation, m._opacity)],
                     pkg["viewspace_points"].grad.clone()))
    torch.testing.assert_close(outs[0][0], outs[1][0], rtol=1e-5, atol=1e-6)
    # gradients: sums over pixels with cancellation, and the two paths round the activations differently in the last bit:
    # 2e-4 relative, with an absolute floor of 1e-5 of the largest entry
    for x, y in zip(outs[0][1] + [outs[0][2]], outs[1][1] + [outs[1][2]]):
        torch.testing.assert_close(x, y, rtol=2e-4, atol=1e-5 * float(y.abs().max()))


def test_rasterizer_raw_activations_flag_with_aa_depth_and_scale_modifier():
    """`GaussianRasterizer.forward(..., raw_activations=True)` on the raw parameters against the same rasterizer fed with
    torch.exp / F.normalize / torch.sigmoid of them (autograd through the PyTorch ops): anti-aliasing on (the opacity
    enters the AA factor), scale_modifier != 1, gradient on colour and inverse depth, the dc / rest call form.  Forward
    differences come only from the last-bit rounding of the activations; gradients agree to 2e-4 relative."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from helpers import settings_for, upstream_grads
    from scene_utils.cameras import look_at_camera
    from scene_utils.synthetic import make_gaussians

    raw = make_gaussians(1800, 3, seed=21).to("cuda")
    cam = look_at_camera((0.5, -4.0, 0.8), (0, 0, 0), (0, 0, 1), 1.0, 112, 80).to("cuda")
    bg = torch.tensor([0.2, 0.1, 0.0], device="cuda")
    rs = settings_for(cam, 3, bg, scale_modifier=1.3, antialiasing=True, cls=GaussianRasterizationSettings, device="cuda")
    gc, gd = (t.cuda() for t in upstream_grads(80, 112))
    res = []
    for use_raw in (True, False):
        xyz = raw.xyz.clone().requires_grad_(True)
        s, q, o = (t.clone().requires_grad_(True) for t in (raw.scaling, raw.rotation, raw.opacity))
        dc = raw.features_dc.clone().requires_grad_(True)
        rest = raw.features_rest.clone().requires_grad_(True)
        m2d = torch.zeros_like(xyz, requires_grad=True)
        if use_raw:
            kw = dict(opacities=o, scales=s, rotations=q, raw_activations=True)
        else:
            kw = dict(opacities=torch.sigmoid(o), scales=torch.exp(s), rotations=torch.nn.functional.normalize(q))
        color, radii, invd = GaussianRasterizer(rs)(means3D=xyz, means2D=m2d, dc=dc, shs=rest, **kw)
        ((color * gc).sum() + (invd * gd).sum()).backward()
        res.append((color.detach(), invd.detach(), radii, [t.grad.clone() for t in (xyz, s, q, o, dc, rest, m2d)]))
    (c0, d0, r0, g0), (c1, d1, r1, g1) = res
    assert int((r0 > 0).sum()) > 300
    assert int((r0 != r1).sum()) <= 2
    diff = (c0 - c1).abs().amax(dim=0)
    assert int((diff >= 2e-5).sum()) <= 3 and float(diff.max()) < 5e-3      # at most a few alpha-threshold flips
    assert float((d0 - d1).abs().max()) < 5e-3
    for a, b in zip(g0, g1):
        assert float((a - b).norm() / b.norm().clamp_min(1e-20)) < 2e-4
