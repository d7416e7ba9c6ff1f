"""GPU parity tests proper: the HIP path (through the reference-shaped Python boundary -> C ABI) against the CPU
oracle on identical seeded inputs, and against the committed golden fixtures.

fp32 tolerance (north_star "within a stated fp32 tolerance"), oracle evaluated in float64:
  forward colour / inverse depth : |err| <= 2e-5 on >= 99.99 % of pixels, PSNR >= 80 dB
  radii / n_contrib              : exact on >= 99.99 % of entries (ceil / threshold flips allowed on the rest)
  every gradient tensor          : relative L2 <= 1e-4 and max-abs <= 1e-4 * max|g|
  binning (integer work)         : bit-exact against keys rebuilt from the GPU's own depth bits / tile rects
"""
import math
import os

import numpy as np
import pytest
import torch

from helpers import (run_hip, run_oracle, upstream_grads, rel_l2, lowlevel_forward, settings_for, leaf_inputs)
from scene_utils import make_gaussians, fibonacci_cameras, look_at_camera, make_config

pytestmark = pytest.mark.gpu

FWD_ATOL, FWD_FRAC, GRAD_REL, EXACT_FRAC = 2e-5, 0.9999, 1e-4, 0.9999


def check_forward(out, ref):
    # radii: exact on >= 99.99 % of entries; the rest (at least 2 allowed for small P) may only be a +-1 ceil() flip or a
    # visibility flip at a cull threshold - fp32 vs float64 rounding at a discontinuity
    r0, r1 = ref["radii"].long(), out["radii"].long()
    bad = r0 != r1
    assert int(bad.sum()) <= max(2, int(1e-4 * r0.numel())), int(bad.sum())
    assert bool((((r0 - r1).abs() <= 1) | (r0 == 0) | (r1 == 0))[bad].all())
    for k in ("color", "invdepth"):
        d = (ref[k].double() - out[k].double()).abs()
        assert float((d <= FWD_ATOL).double().mean()) >= FWD_FRAC, (k, float(d.max()))
    mse = float(((ref["color"].double() - out["color"].double()) ** 2).mean())
    assert mse == 0 or 10 * math.log10(1.0 / mse) >= 80.0


def check_grads(out, ref, rel=GRAD_REL, flip_rows=0):
    """The stated bar, no slack: per tensor rel-L2 <= 1e-4 and max-abs <= 1e-4 max|g|.
    `flip_rows` (only test_config1_full_size passes one: max(2, 1e-4 P), the budget the forward check gives radii and pixels):
    rows that may leave the bar because a PIXEL FLIPPED at one of the rasterizer's discontinuities (alpha >= 1/255, power <= 0,
    T < 1e-4) between float32 and float64 - that Gaussian's gradient then gains or loses a whole pixel's contribution, which
    no float32 evaluation can avoid.  Evidence that this is what C1's outliers are (build container, CPU only): the independent
    torch float32 oracle against its own float64 run misses the bar on the same tensors, worst at row 2327 of `scales`
    = (-18.96, -4.13, -0.60) with errors (0.004, 0.30, 0.22) - 7 % of one component, a lost contribution, not rounding noise;
    which rows flip depends on the evaluation order (the float32 oracle's worst element is 1.2e-3 max|g| with 8 CPU threads,
    1.4e-4 with 16; the kernels' is 2.3e-4).  Such rows stay bounded (<= 1e-2 max|g|) and everything else meets the bar."""
    for k, g_ref in ref["grads"].items():
        g = out["grads"][k]
        assert g.shape == g_ref.shape, k
        if g_ref.numel() == 0:          # (SH degree 0 in the dc + rest call form: an empty rest tensor)
            continue
        assert torch.isfinite(g).all(), k
        if float(g_ref.abs().max()) == 0.0:
            assert float(g.abs().max()) == 0.0, k
            continue
        err = (g.double() - g_ref.double()).abs()
        bar = rel * float(g_ref.abs().max())
        if rel_l2(g, g_ref) <= rel and float(err.max()) <= bar:
            continue
        assert flip_rows > 0, (k, rel_l2(g, g_ref), float(err.max()) / float(g_ref.abs().max()))
        over = err.view(err.shape[0], -1).max(dim=1).values > bar
        assert int(over.sum()) <= flip_rows, (k, int(over.sum()))
        assert float(err.max()) <= 1e-2 * float(g_ref.abs().max()), (k, float(err.max()) / float(g_ref.abs().max()))
        assert rel_l2(g[~over], g_ref[~over]) <= rel, (k, rel_l2(g[~over], g_ref[~over]))


def small_scene(P=3000, W=150, H=100, deg=3, seed=11, scale=0.6, view=1):
    raw = make_gaussians(P, deg, seed=seed, scale_factor=scale)
    cam = fibonacci_cameras(3, W, H, seed=5)[view]
    return raw, cam


@pytest.mark.parametrize("mode,aa,cov_precomp", [
    ("sh", False, False), ("sh", True, False), ("dc", False, False), ("colors", False, False),
    ("sh", False, True), ("colors", True, True)])
def test_forward_backward_parity(mode, aa, cov_precomp):
    raw, cam = small_scene()
    bg = torch.tensor([0.2, 0.5, 0.7])
    gc, gd = upstream_grads(cam.image_height, cam.image_width)
    ref = run_oracle(raw, cam, 3, bg, torch.float64, mode=mode, antialiasing=aa, gc=gc, gd=gd, cov_precomp=cov_precomp)
    out = run_hip(raw, cam, 3, bg, mode=mode, antialiasing=aa, gc=gc, gd=gd, cov_precomp=cov_precomp, debug=True)
    check_forward(out, ref)
    check_grads(out, ref)
    # screen-space gradient side channel: z component is zero
    assert float(out["grads"]["means2D"][:, 2].abs().max()) == 0.0


@pytest.mark.parametrize("aa", [0, 1])
def test_against_committed_golden(aa):
    """No oracle call: inputs and expected outputs come from tests/golden/oracle_small_scene.npz."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_small_scene.npz"))
    dev = "cuda"
    t = {k[3:]: torch.tensor(G[k], dtype=torch.float32, device=dev).requires_grad_(True) for k in G.files if k.startswith("in_")}
    H, W = G["gc"].shape[1:]
    m2d = torch.zeros(t["means3D"].shape[0], 3, device=dev, requires_grad=True)
    s = GaussianRasterizationSettings(H, W, math.tan(G["fov"][0] * 0.5), math.tan(G["fov"][1] * 0.5),
                                      torch.tensor(G["bg"], device=dev), 1.0, torch.tensor(G["viewmatrix"], device=dev),
                                      torch.tensor(G["projmatrix"], device=dev), 3, torch.tensor(G["campos"], device=dev),
                                      False, False, bool(aa))
    color, radii, invd = GaussianRasterizer(s)(t["means3D"], m2d, t["opacities"], shs=t["shs"], scales=t["scales"],
                                               rotations=t["rotations"])
    ((color * torch.tensor(G["gc"], device=dev)).sum() + (invd * torch.tensor(G["gd"], device=dev)).sum()).backward()
    out = dict(color=color.detach().cpu(), invdepth=invd.detach().cpu(), radii=radii.cpu(),
               grads={k: v.grad.cpu() for k, v in t.items()})
    out["grads"]["means2D"] = m2d.grad.cpu()
    ref = dict(color=torch.tensor(G[f"aa{aa}_color"]), invdepth=torch.tensor(G[f"aa{aa}_invdepth"]),
               radii=torch.tensor(G[f"aa{aa}_radii"]),
               grads={k: torch.tensor(G[f"aa{aa}_grad_{k}"]) for k in list(t) + ["means2D"]})
    check_forward(out, ref)
    check_grads(out, ref)


def test_binning_bit_exact_and_image_state():
    raw, cam = small_scene(P=5000, W=200, H=120)
    bg = torch.zeros(3)
    W, H = cam.image_width, cam.image_height
    ll = lowlevel_forward(raw, cam, 3, bg)
    P = 5000
    tt, rect, order = ll["tiles_touched"], ll["rect"].astype(np.int64), ll["order"]
    depth_bits = ll["rec"][:, 11].numpy().view(np.uint32).astype(np.uint64)
    vis = ll["radii"].numpy() > 0
    assert not (tt[~vis] > 0).any()          # culled Gaussians emit nothing; visible ones may emit 0 tiles (exact tile culling)
    # depth order: sorted keys ascending, permutation of all ids, stable on ties
    ks = ll["depth_keys_sorted"]
    assert (np.diff(ks.astype(np.int64)) >= 0).all()
    assert sorted(order.tolist()) == list(range(P))
    exp_keys = np.where(vis, depth_bits.astype(np.uint32), np.uint32(0xFFFFFFFF))
    assert (np.argsort(exp_keys, kind="stable") == order).all()
    # inclusive scan in depth order
    assert (np.cumsum(tt[order].astype(np.int64)) == ll["offsets"].astype(np.int64)).all()
    assert int(ll["offsets"][-1]) == ll["R"]
    # instance list.  Inside `rect` a tile is emitted only if the alpha >= 1/255 ellipse can reach it, so the expected list is
    # rebuilt from the emitted (tile, Gaussian) pairs themselves: they must be distinct, lie inside rect, number
    # tiles_touched per Gaussian, and be ordered exactly like a stable sort of (tile << 32 | depth bits) taken in
    # ascending-id emission order (ties keep ascending Gaussian id).
    gx = (W + 15) // 16
    pl, rg = ll["point_list"].astype(np.int64), ll["ranges"].astype(np.int64)
    assert len(pl) == ll["R"] == int(tt.sum())
    tile_of_pos = np.full(len(pl), -1, dtype=np.int64)
    for t in range(rg.shape[0]):
        tile_of_pos[rg[t, 0]:rg[t, 1]] = t
    assert (tile_of_pos >= 0).all()                                            # ranges tile the list completely
    assert (np.diff(rg[rg[:, 1] > rg[:, 0]].reshape(-1)) >= 0).all()          # and in tile order
    pair = tile_of_pos * P + pl
    assert len(np.unique(pair)) == len(pair)                                  # no duplicates
    tyx = np.stack([tile_of_pos % gx, tile_of_pos // gx], 1)
    assert ((tyx[:, 0] >= rect[pl, 0]) & (tyx[:, 0] < rect[pl, 2]) & (tyx[:, 1] >= rect[pl, 1]) & (tyx[:, 1] < rect[pl, 3])).all()
    assert (np.bincount(pl, minlength=P) == tt).all()
    key = (tile_of_pos.astype(np.uint64) << np.uint64(32)) | depth_bits[pl]
    order_key = key.astype(object) * (2 ** 32) + pl                            # (tile, depth bits, id) as one integer
    assert all(order_key[i] < order_key[i + 1] for i in range(len(order_key) - 1))
    assert int(ll["ranges"][:, 1].max()) == ll["R"]
    # image state vs oracle.  The instance lists are shorter than the oracle's (tiles that provably cannot contribute are
    # not emitted), so list POSITIONS differ; what must agree is WHICH Gaussian is each pixel's last contributor, and T.
    ref = run_oracle(raw, cam, 3, bg, torch.float64)
    st = ref["state"]
    assert ll["R"] <= len(st["point_list"])
    gyx = np.arange(H)[:, None] // 16 * gx + np.arange(W)[None, :] // 16          # tile of every pixel

    def last_id(n_contrib, ranges, plist):
        n = np.asarray(n_contrib, dtype=np.int64)
        pos = np.asarray(ranges)[gyx, 0].astype(np.int64) + n - 1
        ids = np.asarray(plist, dtype=np.int64)[np.clip(pos, 0, max(len(plist) - 1, 0))]
        return np.where(n > 0, ids, -1)
    a = last_id(ll["n_contrib"].numpy(), ll["ranges"], ll["point_list"])
    b = last_id(st["n_contrib"].numpy(), st["ranges"].numpy(), st["point_list"].numpy())
    assert (a == b).mean() >= EXACT_FRAC
    assert float(((ll["final_T"].double() - st["final_T"]).abs() <= 2e-5).double().mean()) >= FWD_FRAC
    # every emitted instance lies inside the published 3-sigma rectangle of its Gaussian
    pre = st["pre"]
    rmin, rmax = pre.rect_min.numpy(), pre.rect_max.numpy()
    sub = rect[tt > 0]
    assert (sub[:, 0] >= rmin[tt > 0, 0]).all() and (sub[:, 2] <= rmax[tt > 0, 0]).all()
    assert (sub[:, 1] >= rmin[tt > 0, 1]).all() and (sub[:, 3] <= rmax[tt > 0, 1]).all()


def test_config1_full_size():
    """BASELINE configs[0]: 10k Gaussians, SH degree 0, 256x256 - the oracle's own CPU-runnable case."""
    raw, cams, c = make_config(1)
    bg = torch.zeros(3)
    gc, gd = upstream_grads(c["H"], c["W"], depth=False)
    ref = run_oracle(raw, cams[0], 0, bg, torch.float64, gc=gc, gd=gd)
    out = run_hip(raw, cams[0], 0, bg, gc=gc, gd=gd)
    check_forward(out, ref)
    check_grads(out, ref, flip_rows=max(2, int(1e-4 * c["P"])))


@pytest.mark.parametrize("depth,aa,aniso", [(False, False, 0.0), (True, True, 0.0), (False, False, 1.3)])
def test_backward_subblock_masks_change_no_bit(depth, aa, aniso):
    """Round 4: the one-wave-per-tile compositing backward skips, on the scalar unit, the 8x8 sub-blocks of a tile a Gaussian
    cannot reach (masks computed where its records are staged, csrc/render.hip gsr_subblock_mask).  A skipped sub-block would only
    have added exact zeros, so every gradient must equal the unmasked loop's (GSR_BWD_MASK=0: the round-3 kernel) BIT FOR BIT -
    also with the inverse-depth gradient + anti-aliasing instantiation and on a scene of needles and pancakes (per-axis log-scale
    spread 1.3: the conservative rectangle test against strongly correlated conics).  The regular scenes also agree with the float64 oracle."""
    raw = make_gaussians(6000, 2, seed=401, scale_factor=0.8)
    if aniso:
        gen = torch.Generator().manual_seed(402)
        raw.scaling = raw.scaling + aniso * torch.randn(raw.scaling.shape, generator=gen)
    cam = fibonacci_cameras(3, 208, 144, seed=403)[2]
    bg = torch.tensor([0.3, 0.2, 0.1])
    gc, gd = upstream_grads(cam.image_height, cam.image_width, depth=depth)
    old = {k: os.environ.get(k) for k in ("GSR_BWD_FORM", "GSR_BWD_MASK", "GSR_BWD_REDUCE")}
    try:
        os.environ["GSR_BWD_FORM"] = "tile"
        os.environ["GSR_BWD_REDUCE"] = "swap"       # (the default: the halving-tree form, which shares its tree with round 3)
        os.environ["GSR_BWD_MASK"] = "0"
        a = run_hip(raw, cam, 2, bg, antialiasing=aa, gc=gc, gd=gd if depth else None)
        os.environ["GSR_BWD_MASK"] = "1"
        b = run_hip(raw, cam, 2, bg, antialiasing=aa, gc=gc, gd=gd if depth else None)
        os.environ["GSR_BWD_REDUCE"] = "mfma"       # opt-in form: the same masked walk, sums on the matrix pipe (k_render_bwd_tile_mx)
        m = run_hip(raw, cam, 2, bg, antialiasing=aa, gc=gc, gd=gd if depth else None)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert torch.equal(a["color"], b["color"])
    for k in a["grads"]:
        assert torch.equal(a["grads"][k], b["grads"][k]), (k, float((a["grads"][k] - b["grads"][k]).abs().max()))
    assert float(b["grads"]["means3D"].abs().sum()) > 0
    # the matrix-pipe form adds the same terms in another (fixed) order: equal to fp32 rounding, not bit for bit
    ref = run_oracle(raw, cam, 2, bg, torch.float64, antialiasing=aa, gc=gc, gd=gd)     # (depth=False: gd is all zeros)
    for k in b["grads"]:
        gb, gm, gr = b["grads"][k].double().cpu(), m["grads"][k].double().cpu(), ref["grads"][k].double()
        if not aniso:
            assert float((gb - gm).norm()) <= 2e-6 * float(gb.norm()) + 1e-30, (k, float((gb - gm).norm() / gb.norm()))
            assert float((gb - gm).abs().max()) <= 1e-5 * float(gb.abs().max()) + 1e-30, (k, float((gb - gm).abs().max()))
        else:
            # needles: float32 itself is 1e-3 .. 1e-2 away from float64 on this scene (ill-conditioned conics; measured for both
            # forms, tests/sweeps/mx_vs_swap_error.py), so the two orders of summation differ by that much from each other; what
            # is asked of the matrix-pipe form is that it is no farther from float64 than the tree
            eb, em = float((gb - gr).norm() / gr.norm()), float((gm - gr).norm() / gr.norm())
            assert em <= 1.5 * eb + 2e-6, (k, em, eb)
    if not aniso:      # (the needle scene is there for the masks; its conics are too ill-conditioned for the 2e-5 image bar)
        check_forward(b, ref)
        check_grads(b, ref)
        check_grads(m, ref)


@pytest.mark.parametrize("aa,aniso,P,scale", [(False, 0.0, 6000, 0.8), (True, 1.3, 6000, 0.8), (False, 0.0, 20000, 2.5)])
def test_forward_quadrant_masks_change_no_bit(aa, aniso, P, scale):
    """Round 4, opt-in variant (GSR_FWD_MASK=1; measured slower than the plain loop, kept as evidence and exercised here): the
    compositing forward walks, per 8x8 quadrant wave, only the list entries whose alpha >= 1/255 ellipse can reach that quadrant
    (masks computed by the staging threads).  Entries it does walk take the published test in the published order, so
    colour, inverse depth and - through final_T / n_contrib, which the backward replays - every gradient must equal the unmasked
    loop's (GSR_FWD_MASK=0: the round-3 kernel) bit for bit; also on needles (anisotropic scales) and on a saturating scene (big
    splats: most pixels finish early, the early-out and the masks interact)."""
    raw = make_gaussians(P, 2, seed=411, scale_factor=scale)
    if aniso:
        gen = torch.Generator().manual_seed(412)
        raw.scaling = raw.scaling + aniso * torch.randn(raw.scaling.shape, generator=gen)
    cam = fibonacci_cameras(3, 208, 144, seed=413)[1]
    bg = torch.tensor([0.3, 0.2, 0.1])
    gc, gd = upstream_grads(cam.image_height, cam.image_width)
    old = os.environ.get("GSR_FWD_MASK")
    try:
        os.environ["GSR_FWD_MASK"] = "0"
        a = run_hip(raw, cam, 2, bg, antialiasing=aa, gc=gc, gd=gd)
        os.environ["GSR_FWD_MASK"] = "1"
        b = run_hip(raw, cam, 2, bg, antialiasing=aa, gc=gc, gd=gd)
    finally:
        if old is None:
            os.environ.pop("GSR_FWD_MASK", None)
        else:
            os.environ["GSR_FWD_MASK"] = old
    assert torch.equal(a["color"], b["color"]) and torch.equal(a["invdepth"], b["invdepth"]) and torch.equal(a["radii"], b["radii"])
    for k in a["grads"]:
        assert torch.equal(a["grads"][k], b["grads"][k]), (k, float((a["grads"][k] - b["grads"][k]).abs().max()))
    assert float(b["color"].sum()) > 0 and float(b["grads"]["means3D"].abs().sum()) > 0


def test_bitwise_reproducible():
    """No float atomics anywhere: two runs give bit-identical images AND gradients (the reference's atomics do not)."""
    raw, cam = small_scene()
    bg = torch.tensor([0.1, 0.1, 0.1])
    gc, gd = upstream_grads(cam.image_height, cam.image_width)
    a = run_hip(raw, cam, 3, bg, gc=gc, gd=gd)
    b = run_hip(raw, cam, 3, bg, gc=gc, gd=gd)
    assert torch.equal(a["color"], b["color"]) and torch.equal(a["invdepth"], b["invdepth"])
    for k in a["grads"]:
        assert torch.equal(a["grads"][k], b["grads"][k]), k


def test_edge_cases():
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C
    dev = "cuda"
    bg = torch.tensor([0.3, 0.6, 0.9])
    # (1) P = 0 -> background, no launch problems
    cam = fibonacci_cameras(1, 40, 24, seed=1)[0]
    s = settings_for(cam, 0, bg, cls=GaussianRasterizationSettings, device=dev)
    e = torch.zeros(0, 3, device=dev)
    color, radii, invd = GaussianRasterizer(s)(e, e, torch.zeros(0, 1, device=dev), shs=torch.zeros(0, 1, 3, device=dev),
                                               scales=e, rotations=torch.zeros(0, 4, device=dev))
    assert torch.allclose(color.cpu(), bg[:, None, None].expand(3, 24, 40)) and radii.numel() == 0
    assert float(invd.abs().max()) == 0
    # (2) everything behind the camera / far off-screen -> background, zero gradients
    raw = make_gaussians(64, 1, seed=3)
    raw.xyz[:] = torch.tensor([0.0, 0.0, 0.0]) + 40.0
    gc, gd = upstream_grads(24, 40)
    out = run_hip(raw, cam, 1, bg, gc=gc, gd=gd)
    assert int((out["radii"] > 0).sum()) == 0
    assert torch.allclose(out["color"], bg[:, None, None].expand(3, 24, 40))
    for k, g in out["grads"].items():
        assert float(g.abs().max()) == 0.0, k
    # (3) image smaller than one tile, single huge Gaussian covering it, opacity ~1 (alpha clamps at 0.99)
    cam2 = look_at_camera((0, -3.0, 0), (0, 0, 0), (0, 0, 1), 0.8, 10, 7)
    one = make_gaussians(1, 0, seed=4)
    one.xyz[:] = 0; one.scaling[:] = math.log(2.0); one.opacity[:] = 9.0
    gc, gd = upstream_grads(7, 10)
    ref = run_oracle(one, cam2, 0, bg, torch.float64, gc=gc, gd=gd)
    out = run_hip(one, cam2, 0, bg, gc=gc, gd=gd)
    assert int(out["radii"][0]) > 0
    check_forward(out, ref); check_grads(out, ref)
    # (4) prefiltered=True with a culled point is a hard error (SURVEY 8b error conventions)
    s = settings_for(cam, 1, bg, cls=GaussianRasterizationSettings, device=dev)._replace(prefiltered=True)
    inp = leaf_inputs(raw, torch.float32, dev)
    behind = inp["means3D"].detach().clone(); behind[:] = cam.camera_center.to(dev) * 1.5   # behind the camera
    with pytest.raises(_C.GsrError, match="prefiltered"):
        GaussianRasterizer(s)(behind, inp["means2D"], inp["opacities"], shs=inp["shs"], scales=inp["scales"],
                              rotations=inp["rotations"])


def test_frustum_clamp_and_scale_modifier_and_low_active_degree():
    """Gaussians beyond 1.3*tanfov (clamp masks, quirk q2), scale_modifier != 1, active SH degree below the stored one."""
    raw = make_gaussians(1500, 3, seed=31, scale_factor=2.5, box=3.5)       # wide box: many outside the frustum
    cam = look_at_camera((0.5, -2.2, 0.4), (0, 0, 0), (0, 0, 1), 0.5, 96, 64)
    bg = torch.tensor([0.0, 0.2, 0.0])
    gc, gd = upstream_grads(64, 96)
    for deg, mod in ((3, 1.0), (1, 0.7), (0, 1.3)):
        ref = run_oracle(raw, cam, deg, bg, torch.float64, gc=gc, gd=gd, scale_modifier=mod)
        out = run_hip(raw, cam, deg, bg, gc=gc, gd=gd, scale_modifier=mod)
        pre = ref["state"]["pre"]
        check_forward(out, ref); check_grads(out, ref)
        if deg < 3:       # stored-but-inactive SH bands receive exactly zero gradient
            K = (deg + 1) ** 2
            assert float(out["grads"]["shs"][:, K:].abs().max()) == 0.0
    # the scene really exercises the clamp region
    s = settings_for(cam, 3, bg)
    t = raw.xyz.double() @ s.viewmatrix.double()[:3, :3] + s.viewmatrix.double()[3, :3]
    vis = ref["radii"] > 0
    outside = ((t[:, 0] / t[:, 2]).abs() > 1.3 * s.tanfovx) | ((t[:, 1] / t[:, 2]).abs() > 1.3 * s.tanfovy)
    assert int((vis & outside).sum()) >= 5


def test_mark_visible_and_no_grad_forward():
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from oracle import gs_oracle as O
    raw, cam = small_scene(P=2000)
    dev = "cuda"
    s = settings_for(cam, 3, torch.zeros(3), cls=GaussianRasterizationSettings, device=dev)
    r = GaussianRasterizer(s)
    xyz = (raw.xyz * 3).to(dev)
    vis = r.markVisible(xyz).cpu()
    assert vis.dtype == torch.bool and (vis == O.mark_visible(raw.xyz * 3, cam.world_view_transform)).all()
    inp = leaf_inputs(raw, torch.float32, dev)
    with torch.no_grad():
        color, radii, invd = r(inp["means3D"], inp["means2D"], inp["opacities"], shs=inp["shs"], scales=inp["scales"],
                               rotations=inp["rotations"])
    assert not color.requires_grad
    ref = run_oracle(raw, cam, 3, torch.zeros(3), torch.float64)
    check_forward(dict(color=color.cpu(), invdepth=invd.cpu(), radii=radii.cpu()), ref)


def test_render_boundary_contract():
    """render(): returned keys, viewspace_points.grad, python-vs-native SH / covariance branches agree
    (reference gaussian_renderer/__init__.py:60-86,118-121)."""
    from gaussian_renderer import render, PipelineParams
    from scene_utils import GaussianModel
    raw, cam = small_scene(P=2500)
    cam.to("cuda")
    model = GaussianModel.from_raw(raw.to("cuda"))
    bg = torch.tensor([0.5, 0.5, 0.5], device="cuda")
    imgs = {}
    for name, pipe, kw in (("native", PipelineParams(), {}),
                           ("py_sh", PipelineParams(convert_SHs_python=True), {}),
                           ("py_cov", PipelineParams(compute_cov3D_python=True), {}),
                           ("sep", PipelineParams(), dict(separate_sh=True))):
        pkg = render(cam, model, pipe, bg, **kw)
        assert set(pkg) >= {"render", "viewspace_points", "visibility_filter", "radii"}
        assert pkg["render"].shape == (3, cam.image_height, cam.image_width)
        assert pkg["radii"].dtype == torch.int32 and pkg["visibility_filter"].dtype == torch.bool
        pkg["render"].sum().backward()
        g = pkg["viewspace_points"].grad
        assert g is not None and g.shape == (2500, 3) and float(g[:, 2].abs().max()) == 0
        assert float(g[pkg["visibility_filter"]].abs().sum()) > 0
        assert float(g[~pkg["visibility_filter"]].abs().sum()) == 0
        grads = [p.grad.clone() for p in model.parameters()]
        for p in model.parameters():
            p.grad = None
        imgs[name] = (pkg["render"].detach(), grads)
    for name in ("py_sh", "py_cov", "sep"):
        # the branches round covariance / colour differently in the last bit: all pixels agree to 2e-5 except possibly a few
        # where one (pixel, Gaussian) pair sits exactly on the alpha >= 1/255 cut-off and flips (a step of <= 1/255)
        d = (imgs[name][0] - imgs["native"][0]).abs().amax(dim=0)
        assert int((d >= 2e-5).sum()) <= 3 and float(d.max()) < 5e-3, (name, int((d >= 2e-5).sum()), float(d.max()))
        for ga, gb in zip(imgs[name][1], imgs["native"][1]):
            assert rel_l2(ga.cpu(), gb.cpu()) < 2e-4, name


def test_full_size_1080p_sampled_tiles_and_properties():
    """BASELINE configs[1] (100k Gaussians, SH3, 1920x1080): the oracle renders 48 sampled tiles (fwd+bwd); the HIP path
    renders the full frame with dL/dpixel masked to those tiles.  Plus size-independent properties on the full frame."""
    raw, cams, c = make_config(2)
    cam = cams[0]
    W, H = c["W"], c["H"]
    gx = (W + 15) // 16
    bg = torch.tensor([0.05, 0.05, 0.05])
    gen = torch.Generator().manual_seed(9)
    tiles = sorted(torch.randperm(gx * ((H + 15) // 16), generator=gen)[:48].tolist())
    mask = torch.zeros(1, H, W)
    for t in tiles:
        ty, tx = divmod(t, gx)
        mask[:, ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16] = 1
    gc, gd = upstream_grads(H, W)
    gc, gd = gc * mask, gd * mask
    ref = run_oracle(raw, cam, 3, bg, torch.float64, gc=gc, gd=gd, tiles=tiles)
    out = run_hip(raw, cam, 3, bg, gc=gc, gd=gd)
    sel = mask.bool()
    d = (ref["color"].double() - out["color"].double()).abs()[sel.expand(3, H, W)]
    assert float((d <= FWD_ATOL).double().mean()) >= FWD_FRAC
    assert float((ref["radii"] == out["radii"]).float().mean()) >= EXACT_FRAC
    check_grads(out, ref)
    # properties: linearity of the backward in the upstream gradient, bg-only where nothing lands
    out2 = run_hip(raw, cam, 3, bg, gc=2.0 * gc, gd=2.0 * gd)
    for k in out["grads"]:
        assert rel_l2(out2["grads"][k], 2.0 * out["grads"][k]) < 1e-5, k
    ll = lowlevel_forward(raw, cam, 3, bg)
    empty = ll["n_contrib"] == 0
    assert torch.allclose(ll["color"][:, empty], bg[:, None].expand(3, int(empty.sum())))
    assert float(ll["final_T"].min()) >= 0 and float(ll["final_T"].max()) <= 1
    assert int(ll["ranges"][:, 1].max()) == ll["R"]


def test_wave_reduction_primitive():
    """The recursive-halving cross-lane reduction of the render backward (v_permlane32/16_swap + DPP) against exact sums:
    integer-valued floats make every partial sum exact, so any lane-mapping mistake shows as a wrong integer."""
    from diff_gaussian_rasterization import _C
    lib = _C.lib()
    gen = torch.Generator().manual_seed(4)
    for trial in range(3):
        x = torch.randint(-1000, 1000, (10, 64), generator=gen).float()
        if trial == 0:
            x = (torch.arange(640).view(10, 64) % 97).float() * (torch.arange(10).view(10, 1) + 1)
        xin = x.cuda().contiguous()
        for hook in (lib.gsr_debug_wave_reduce, lib.gsr_debug_wave_reduce_pk):      # scalar trees; packed-pair trees (round 4)
            out = torch.zeros(20, device="cuda")
            _C.check(hook(_C.ptr(xin), _C.ptr(out), _C._stream()))
            torch.cuda.synchronize()
            assert torch.equal(out.cpu()[:10], x.sum(dim=1)), (out.cpu(), x.sum(dim=1))        # ten-value tree
            assert torch.equal(out.cpu()[10:19], x.sum(dim=1)[:9]), (out.cpu(), x.sum(dim=1))  # nine-value tree
    # and on arbitrary floats the packed trees give the bits of the scalar ones (the same summation tree per value)
    x = torch.randn(10, 64, generator=gen).cuda().contiguous()
    o1, o2 = torch.zeros(20, device="cuda"), torch.zeros(20, device="cuda")
    _C.check(lib.gsr_debug_wave_reduce(_C.ptr(x), _C.ptr(o1), _C._stream()))
    _C.check(lib.gsr_debug_wave_reduce_pk(_C.ptr(x), _C.ptr(o2), _C._stream()))
    torch.cuda.synchronize()
    assert torch.equal(o1, o2)


def _walk_class(w):
    """gsr_walk_class of csrc/gsr_common.h (4 classes per octave): exponent and two leading mantissa bits, clamped at 65535."""
    w = min(int(w), 65535)
    if w < 4:
        return w
    e = w.bit_length() - 1
    return (e << 2) + ((w >> (e - 2)) & 3)


@pytest.mark.parametrize("W,H,P,scale", [(208, 144, 6000, 0.8), (100, 52, 800, 2.5), (64, 48, 3, 1.0)])
def test_walk_classes_of_a_frame(W, H, P, scale):
    """Round 4: a forward with a backward to follow files every tile under the class of its walk length - the deepest contributor
    of any of its pixels, i.e. the number of list entries the compositing backward replays for it - and k_render_bwd_tile takes the
    classes longest first.  Straight from the C ABI (blocking path: the counters are cleared by a memset there): every tile is
    filed exactly once, under the class of max(n_contrib) over its pixels, and walk_of_tile holds that maximum (the workgroup of a
    tile's INDEX writes the zero records behind it).  Odd image sizes, tiles without instances (3 Gaussians), big splats."""
    import numpy as np
    raw = make_gaussians(P, 1, seed=77, scale_factor=scale)
    cam = fibonacci_cameras(2, W, H, seed=78)[1]
    out = lowlevel_forward(raw, cam, 1, torch.tensor([0.1, 0.2, 0.3]))
    gx, gy = (W + 15) // 16, (H + 15) // 16
    tiles = gx * gy
    nc = out["n_contrib"].numpy()
    want = np.zeros(tiles, dtype=np.int64)
    for t in range(tiles):
        ty, tx = divmod(t, gx)
        want[t] = nc[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16].max()
    assert np.array_equal(out["walk_of_tile"].astype(np.int64), want)
    cnt = out["walk_cnt"].astype(np.int64)
    assert int(cnt.sum()) == tiles
    seen = []
    for c in range(len(cnt)):
        members = out["walk_list"][c, :cnt[c]].astype(np.int64)
        assert all(_walk_class(want[t]) == c for t in members), (c, members[:8], want[members[:8]])
        seen.extend(members.tolist())
    assert sorted(seen) == list(range(tiles))
    if P > 100:
        assert int((cnt > 0).sum()) >= 3           # (the scene does spread over several classes)


@pytest.mark.parametrize("form", ["tile", "quad"])
@pytest.mark.parametrize("depth,aa,P,scale", [(False, False, 6000, 0.8), (True, True, 6000, 0.8), (False, False, 40, 3.0)])
def test_walk_order_changes_no_bit(depth, aa, P, scale, form):
    """The order in which the compositing backward (both forms) takes the tiles (walk classes, longest first; the zero records behind a
    tile's walk written by the workgroup of the tile's index) must not enter any result: every gradient equals the index-order
    launch's (GSR_BWD_LPT=0) bit for bit - also with the inverse-depth + anti-aliasing instantiation and on a scene most of whose
    tiles are empty (40 big splats)."""
    raw = make_gaussians(P, 2, seed=411, scale_factor=scale)
    cam = fibonacci_cameras(3, 208, 144, seed=413)[1]
    bg = torch.tensor([0.3, 0.2, 0.1])
    gc, gd = upstream_grads(cam.image_height, cam.image_width, depth=depth)
    old = {k: os.environ.get(k) for k in ("GSR_BWD_FORM", "GSR_BWD_LPT")}
    try:
        os.environ["GSR_BWD_FORM"] = form
        os.environ["GSR_BWD_LPT"] = "0"
        a = run_hip(raw, cam, 2, bg, antialiasing=aa, gc=gc, gd=gd if depth else None)
        os.environ["GSR_BWD_LPT"] = "1"             # (forced: a grid this small is left in index order by default)
        b = run_hip(raw, cam, 2, bg, antialiasing=aa, gc=gc, gd=gd if depth else None)
        c = run_hip(raw, cam, 2, bg, antialiasing=aa, gc=gc, gd=gd if depth else None)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert float(b["grads"]["means3D"].abs().sum()) > 0
    for k in a["grads"]:
        assert torch.equal(a["grads"][k], b["grads"][k]), (k, float((a["grads"][k] - b["grads"][k]).abs().max()))
        assert torch.equal(b["grads"][k], c["grads"][k]), k      # (and run to run: which workgroup takes which tile varies)


@pytest.mark.parametrize("form", ["tile", "quad", "tile-mfma"])
@pytest.mark.parametrize("depth,aa,P,scale", [(False, False, 6000, 0.8), (True, True, 3000, 2.5), (False, False, 40, 6.0)])
def test_validity_flags_of_the_gradient_records_change_nothing(depth, aa, P, scale, form):
    """Round 4: on frames of 2.5 M tile instances and more, an instance behind its tile's walk gets no (all-zero) gradient record
    but a zero validity byte, and the projection backward reads the bytes of a Gaussian's slots - sixteen per load - before it
    touches a record.  Forced on here (gsr_debug_set_flags_min_r(0)) for small scenes whose Gaussians have from one to well over
    sixteen instances (big splats: the second flag chunk) and compared with the zero-record form (threshold 2^32 - 1): every
    gradient equal (torch.equal: a sum that skips a +0.0 may come out as -0.0), in all three compositing-backward kernels."""
    from diff_gaussian_rasterization import _C
    lib = _C.lib()
    raw = make_gaussians(P, 2, seed=431, scale_factor=scale)
    cam = fibonacci_cameras(3, 208, 144, seed=433)[0]
    bg = torch.tensor([0.3, 0.2, 0.1])
    gc, gd = upstream_grads(cam.image_height, cam.image_width, depth=depth)
    old = {k: os.environ.get(k) for k in ("GSR_BWD_FORM", "GSR_BWD_REDUCE", "GSR_BWD_LPT")}
    before = lib.gsr_debug_set_flags_min_r(-1)
    try:
        os.environ["GSR_BWD_FORM"] = form.split("-")[0]
        os.environ["GSR_BWD_LPT"] = "1"
        if form.endswith("mfma"):
            os.environ["GSR_BWD_REDUCE"] = "mfma"
        lib.gsr_debug_set_flags_min_r(0xFFFFFFFF)
        a = run_hip(raw, cam, 2, bg, antialiasing=aa, gc=gc, gd=gd if depth else None)
        lib.gsr_debug_set_flags_min_r(0)
        b = run_hip(raw, cam, 2, bg, antialiasing=aa, gc=gc, gd=gd if depth else None)
    finally:
        lib.gsr_debug_set_flags_min_r(before)
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert float(b["grads"]["means3D"].abs().sum()) > 0
    for k in a["grads"]:
        assert torch.equal(a["grads"][k], b["grads"][k]), (k, float((a["grads"][k] - b["grads"][k]).abs().max()))


def test_matrix_pipe_reduction_primitive():
    """Round 4 (opt-in form, GSR_BWD_REDUCE=mfma; measured slower, kept as evidence): k_render_bwd_tile_mx takes an entry's ten sums over the tile's 256 pixels on the matrix pipe (v_mfma_f32_16x16x4_f32
    against the tile's separable pixel basis, csrc/render.hip).  The hook runs that very stage-1 / stage-2 / LDS slot / record code
    on h[4][64], c[4][64] and a mean; with small integer h and half-integer means every product and partial sum is exact in float32,
    so a wrong lane map, basis entry or slot offset shows as a wrong number (asymmetric random data: no row <-> column swap can hide)."""
    import numpy as np
    from diff_gaussian_rasterization import _C
    lib = _C.lib()
    rng = np.random.default_rng(7)
    lane = np.arange(64)
    for trial in range(6):
        h = rng.integers(-8, 9, size=(4, 64)).astype(np.float64)
        c = rng.integers(-100, 101, size=(4, 64)).astype(np.float64)
        if trial == 0:                      # one pixel only: the basis values themselves
            h[:] = 0
            h[3, 37] = 1.0
        if trial == 1:                      # one sub-block only
            h[[0, 1, 3]] = 0
        mu = rng.integers(-20, 21, size=2) * 0.5
        X = np.stack([(lane & 7) + 8 * (s & 1) - 7.5 for s in range(4)])
        Y = np.stack([(lane >> 3) + 8 * (s >> 1) - 7.5 for s in range(4)])
        dx, dy = mu[0] - X, mu[1] - Y
        want = np.array([(h * dx).sum(), (h * dy).sum(), (h * dx * dx).sum(), (h * dx * dy).sum(), (h * dy * dy).sum(), h.sum(),
                         c[0].sum(), c[1].sum(), c[2].sum(), c[3].sum()])
        xin = torch.from_numpy(np.concatenate([h.ravel(), c.ravel(), mu]).astype(np.float32)).cuda().contiguous()
        out = torch.zeros(10, device="cuda")
        _C.check(lib.gsr_debug_mx_reduce(_C.ptr(xin), _C.ptr(out), _C._stream()))
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().astype(np.float64), want), (trial, out.cpu().numpy(), want)


def test_config4_code_path_small():
    """BASELINE configs[3] code path (anti-aliasing + inverse-depth gradient, tile grid wider than 8 bits' worth of tiles
    per row like 4K: 3840/16 = 240 columns) at a size the oracle handles: a 3840x32 strip."""
    raw = make_gaussians(2500, 3, seed=61, scale_factor=1.2)
    raw.xyz[:, 2] *= 0.05                                      # flatten the cloud so the wide strip is well covered
    cam = look_at_camera((0.0, -3.0, 0.0), (0, 0, 0), (0, 0, 1), 1.9, 3840, 32)
    bg = torch.tensor([0.1, 0.0, 0.2])
    gc, gd = upstream_grads(32, 3840)
    ref = run_oracle(raw, cam, 3, bg, torch.float64, antialiasing=True, gc=gc, gd=gd)
    out = run_hip(raw, cam, 3, bg, antialiasing=True, gc=gc, gd=gd)
    assert int((ref["radii"] > 0).sum()) > 500
    check_forward(out, ref)
    check_grads(out, ref)


def test_forward_only_render_is_bit_identical_to_training_forward():
    """Under torch.no_grad() the tile sort carries the Gaussian ids alone (no emission slots for a backward that cannot
    follow): same images, radii and inverse depth, bit for bit, as the gradient-enabled forward."""
    from gaussian_renderer import render, PipelineParams
    from scene_utils import GaussianModel
    raw, cam = small_scene(P=3000, W=200, H=120)
    cam.to("cuda")
    model = GaussianModel.from_raw(raw.to("cuda"))
    bg = torch.tensor([0.2, 0.4, 0.1], device="cuda")
    for aa in (False, True):
        pipe = PipelineParams(antialiasing=aa)
        a = render(cam, model, pipe, bg, separate_sh=True)
        with torch.no_grad():
            b = render(cam, model, pipe, bg, separate_sh=True)
        assert not b["render"].requires_grad
        assert torch.equal(a["render"].detach(), b["render"]) and torch.equal(a["depth"].detach(), b["depth"])
        assert torch.equal(a["radii"], b["radii"])


def test_low_level_C_call_forms():
    """`_C.rasterize_gaussians` / `_C.rasterize_gaussians_backward` / `_C.mark_visible` (call forms of the published extension,
    SURVEY 8b) give the same numbers as the GaussianRasterizer path."""
    from diff_gaussian_rasterization import _C
    raw, cam = small_scene(P=1200, W=96, H=64)
    bg = torch.tensor([0.3, 0.3, 0.1])
    gc, gd = upstream_grads(64, 96)
    ref = run_hip(raw, cam, 3, bg, gc=gc, gd=gd)
    inp = leaf_inputs(raw, torch.float32, "cuda")
    e = torch.empty(0, device="cuda")
    R, color, radii, geom, binning, img, invd = _C.rasterize_gaussians(
        bg.cuda(), inp["means3D"], e, inp["opacities"], inp["scales"], inp["rotations"], 1.0, e,
        cam.world_view_transform.cuda(), cam.full_proj_transform.cuda(), math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
        64, 96, inp["shs"], 3, cam.camera_center.cuda(), False, False, False)
    assert R > 0 and torch.equal(color.cpu(), ref["color"]) and torch.equal(radii.cpu(), ref["radii"])
    grads = _C.rasterize_gaussians_backward(
        bg.cuda(), inp["means3D"], radii, e, inp["opacities"], inp["scales"], inp["rotations"], 1.0, e,
        cam.world_view_transform.cuda(), cam.full_proj_transform.cuda(), math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
        gc.cuda(), gd.cuda(), inp["shs"], 3, cam.camera_center.cuda(), geom, R, binning, img, False, False)
    d_m2, d_col, d_op, d_m3, d_cov, d_sh, d_sc, d_ro = grads
    assert torch.equal(d_m3.cpu(), ref["grads"]["means3D"]) and torch.equal(d_sh.cpu(), ref["grads"]["shs"])
    assert torch.equal(d_m2.cpu(), ref["grads"]["means2D"]) and d_col.numel() == 0 and d_cov.numel() == 0
    # (opacities / scales / rotations reach the leaves through the same tensors here, so they are comparable directly)
    assert torch.equal(d_op.cpu(), ref["grads"]["opacities"]) and torch.equal(d_sc.cpu(), ref["grads"]["scales"])
    vis = _C.mark_visible(inp["means3D"], cam.world_view_transform.cuda(), cam.full_proj_transform.cuda())
    assert vis.dtype == torch.bool and int(vis.sum()) >= int((radii > 0).sum())
    # An image state whose walk classes do not add up to the tile grid (here: wiped by the caller) is composited backward in
    # index order - same gradients, no out-of-range tile (round 4; forced for this 24-tile image, small grids are index order anyway)
    import ctypes as C
    pw = [C.c_void_p() for _ in range(3)]
    classes = _C.lib().gsr_debug_walk_views(_C.ptr(img), 96, 64, C.byref(pw[0]), C.byref(pw[1]), C.byref(pw[2]))
    off = pw[0].value - img.data_ptr()
    assert int(img[off:off + 4 * classes].view(torch.int32).sum()) == 6 * 4       # (every one of the 24 tiles was filed)
    old = {k: os.environ.get(k) for k in ("GSR_BWD_LPT", "GSR_BWD_FORM")}
    try:
        os.environ["GSR_BWD_LPT"] = "1"
        for form in ("tile", "quad"):
            os.environ["GSR_BWD_FORM"] = form
            outs = []
            for wipe in (False, True):
                if wipe:
                    img[off:off + 4 * classes] = 0
                outs.append(_C.rasterize_gaussians_backward(
                    bg.cuda(), inp["means3D"], radii, e, inp["opacities"], inp["scales"], inp["rotations"], 1.0, e,
                    cam.world_view_transform.cuda(), cam.full_proj_transform.cuda(), math.tan(cam.FoVx * 0.5),
                    math.tan(cam.FoVy * 0.5), gc.cuda(), gd.cuda(), inp["shs"], 3, cam.camera_center.cuda(), geom, R, binning,
                    img, False, False))
            for x, y in zip(*outs):
                assert torch.equal(x, y)
            if form == "tile":          # (the quad form is what the reference run above used at this size)
                continue
            assert torch.equal(outs[0][3].cpu(), ref["grads"]["means3D"])
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("mode", ["sh", "dc"])
def test_deferred_colour_path_is_bit_identical(mode):
    """gsr_forward_prepare_geometry + gsr_forward_shade (used under the data-parallel overlap) against the fused K1 path."""
    import diff_gaussian_rasterization as dgr
    raw, cam = small_scene(P=2500)
    bg = torch.tensor([0.2, 0.1, 0.4])
    gc, gd = upstream_grads(cam.image_height, cam.image_width)
    a = run_hip(raw, cam, 3, bg, mode=mode, gc=gc, gd=gd)
    ev = torch.cuda.Event()
    ev.record()
    b = run_hip(raw, cam, 3, bg, mode=mode, gc=gc, gd=gd, sh_ready_event=ev)        # (per call: `sh_ready_event=`)
    assert torch.equal(a["color"], b["color"]) and torch.equal(a["invdepth"], b["invdepth"]) and torch.equal(a["radii"], b["radii"])
    for k in a["grads"]:
        assert torch.equal(a["grads"][k], b["grads"][k]), k


def test_autograd_usage_patterns():
    """Two renders in one graph, retain_graph double backward, partial requires_grad, non-contiguous SH input."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    raw = make_gaussians(2000, 3, seed=71, scale_factor=0.7)
    cams = fibonacci_cameras(3, 112, 80, seed=72)
    bg = torch.tensor([0.1, 0.2, 0.3])
    dev = "cuda"
    inp = leaf_inputs(raw, torch.float32, dev)
    rs = [settings_for(c, 3, bg, cls=GaussianRasterizationSettings, device=dev) for c in cams[:2]]

    def call(s, shs=None, **over):
        kw = dict(means3D=inp["means3D"], means2D=inp["means2D"], opacities=inp["opacities"], shs=inp["shs"] if shs is None else shs,
                  scales=inp["scales"], rotations=inp["rotations"])
        kw.update(over)
        return GaussianRasterizer(s)(**kw)

    # (1) two views in one graph: gradients add up
    c0, _, _ = call(rs[0])
    c1, _, _ = call(rs[1])
    (c0.sum() + 2.0 * c1.sum()).backward()
    both = {k: v.grad.clone() for k, v in inp.items()}
    for v in inp.values():
        v.grad = None
    call(rs[0])[0].sum().backward()
    g0 = {k: v.grad.clone() for k, v in inp.items()}
    for v in inp.values():
        v.grad = None
    (2.0 * call(rs[1])[0].sum()).backward()
    for k, v in inp.items():
        assert rel_l2((g0[k] + v.grad).cpu(), both[k].cpu()) < 1e-6, k
        v.grad = None
    # (2) retain_graph: the saved state survives a first backward
    c0, _, d0 = call(rs[0])
    loss = c0.sum() + d0.sum()
    loss.backward(retain_graph=True)
    first = inp["means3D"].grad.clone()
    inp["means3D"].grad = None
    loss.backward()
    assert torch.equal(first, inp["means3D"].grad)
    for v in inp.values():
        v.grad = None
    # (3) only some inputs require grad
    frozen = inp["shs"].detach()
    c0, _, _ = call(rs[0], shs=frozen)
    c0.sum().backward()
    assert inp["means3D"].grad is not None and frozen.grad is None
    assert rel_l2(inp["means3D"].grad.cpu(), g0["means3D"].cpu()) < 1e-6
    for v in inp.values():
        v.grad = None
    # (4) non-contiguous SH tensor (a transposed view, as the reference builds for its python SH path)
    sh_t = inp["shs"].detach().transpose(1, 2).contiguous().requires_grad_(True)          # [P,3,16]
    c0, _, _ = call(rs[0], shs=sh_t.transpose(1, 2))
    c0.sum().backward()
    assert rel_l2(sh_t.grad.transpose(1, 2).cpu(), g0["shs"].cpu()) < 1e-6


def test_python_sh_branch_at_degree_4():
    """SH degree 4 exists only on the reference's python-SH branch (gaussian_renderer/__init__.py:74-79 with
    --convert_SHs_python, utils/sh_utils.py:102-112; the native kernels stop at degree 3 upstream and here).  render() with
    convert_SHs_python on a degree-4 model against the oracle fed the same precomputed colours, gradients into the 25
    coefficients included."""
    from gaussian_renderer import render, PipelineParams
    from scene_utils import GaussianModel
    from scene_utils.sh import eval_sh
    from oracle import gs_oracle as O
    raw = make_gaussians(1500, 4, seed=401, scale_factor=0.8)
    assert raw.features_rest.shape[1] == 24
    cam = fibonacci_cameras(2, 128, 80, seed=402)[1]
    bg = torch.tensor([0.2, 0.1, 0.3])
    # HIP path through render()
    cam_d = fibonacci_cameras(2, 128, 80, seed=402, device="cuda")[1]
    model = GaussianModel.from_raw(raw.to("cuda"))
    assert model.active_sh_degree == 4
    pkg = render(cam_d, model, PipelineParams(convert_SHs_python=True), bg.cuda())
    gc, gd = upstream_grads(80, 128)
    (pkg["render"] * gc.cuda()).sum().backward()
    # oracle with the colours of the same formula in float64
    act = raw.activated()
    inp = {k: act[k].double().clone().requires_grad_(True) for k in ("means3D", "opacities", "scales", "rotations")}
    fdc = raw.features_dc.double().clone().requires_grad_(True)
    frest = raw.features_rest.double().clone().requires_grad_(True)
    shs = torch.cat((fdc, frest), 1)
    d = inp["means3D"] - cam.camera_center.double()
    col = torch.clamp_min(eval_sh(4, shs.transpose(1, 2), d / d.norm(dim=1, keepdim=True)) + 0.5, 0.0)
    s = settings_for(cam, 4, bg)
    m2d = torch.zeros(1500, 3, dtype=torch.float64, requires_grad=True)
    color, radii, invd = O.rasterize(inp["means3D"], m2d, inp["opacities"], s, colors_precomp=col, scales=inp["scales"],
                                     rotations=inp["rotations"])
    (color * gc.double()).sum().backward()
    assert float((color.detach() - pkg["render"].detach().cpu().double()).abs().max()) < 2e-5
    assert rel_l2(model._features_rest.grad.cpu(), frest.grad) < 1e-4 and rel_l2(model._features_dc.grad.cpu(), fdc.grad) < 1e-4
    assert float(model._features_rest.grad[:, 15:].abs().max()) > 0          # the degree-4 band receives gradient
