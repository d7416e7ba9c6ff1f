"""Tile lists truncated by depth (round 4; include/gsr.h gsr_forward_async_culled, Trainer.enable_tile_cull): every view keeps, per
tile, the depth its slowest pixel saturated at when the view was last rendered (+ a margin); its next unverified render emits only
the instances in front of that.  Exactness rests on two things, both checked here bit for bit:
  * a truncated frame in which every pixel of every truncated tile still saturates inside its list IS the untruncated frame
    (colour, inverse depth, radii, every gradient - the compositing loop never reached the missing tail);
  * a frame for which that does not hold flags itself: its backward - folded optimizer step and statistics included - is a no-op
    on the device, and the Trainer renders the view again untruncated before it touches the next one.
So a training run with truncation must end with the parameters, moments and statistics of the run without."""
import ctypes as C
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _train(cull, steps, scale, kind="hip_fused", shrink_at=None, densify=False):
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _workspace as ws
    from gaussian_renderer import render, PipelineParams
    from scene_utils import make_gaussians, fibonacci_cameras, GaussianModel, Trainer
    dev = "cuda"
    cams = fibonacci_cameras(4, 208, 128, seed=141, device=dev)
    bg = torch.tensor([0.05, 0.1, 0.2], device=dev)
    pipe = PipelineParams()
    teacher = GaussianModel.from_raw(make_gaussians(3000, 2, seed=142, scale_factor=scale).to(dev), requires_grad=False)
    with torch.no_grad():
        gts = {i: render(c, teacher, pipe, bg)["render"].clone() for i, c in enumerate(cams)}
    model = GaussianModel.from_raw(make_gaussians(3000, 2, seed=143, scale_factor=scale).to(dev))
    ws.pool(torch.device(dev, 0)).forget_estimates()
    old = dgr.forward_mode()
    dgr.set_forward_mode("async")
    try:
        tr = Trainer(model, cams, gts, render, pipe, bg, separate_sh=True, optimizer=kind)
        if densify:
            tr.enable_densification(extent=4.4, from_iter=3, until_iter=1000, interval=7, opacity_reset_interval=1000,
                                    grad_threshold=2e-5, min_opacity=0.005, seed=3)
        if cull:
            tr.enable_tile_cull()
        for it in range(steps):
            if cull and shrink_at == it:
                # sabotage: every tile of the view about to be rendered gets a cut-off in front of everything - the frame must flag
                # itself (nothing is emitted, no pixel saturates), be a no-op, and be run again untruncated
                tr._tile_cull_of(it % 4).fill_(0)
            tr.step(it % 4)
        tr.finish()
        torch.cuda.synchronize()
        stats = dgr.call_stats()
    finally:
        dgr.set_forward_mode(old)
    out = []
    for p in model.parameters():
        st = tr.optimizer.state[p]
        out += [p.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone()]
    out += [model.xyz_gradient_accum.clone(), model.denom.clone(), model.max_radii2D.clone()]
    return out, stats, tr


@pytest.mark.parametrize("scale,kind,densify", [(2.2, "hip_fused", False), (0.7, "hip_fused", False), (2.2, "hip_sparse_fused", False),
                                                (2.2, "hip_fused", True)])
def test_training_with_truncated_tile_lists_equals_training_without(scale, kind, densify):
    """24 steps over 4 views (every view comes round 6 times, so its cut-offs are used 5 times): a saturating scene (big splats:
    most tiles cut their lists), a sparse one (background visible: hardly any tile may cut), the sparse optimizer, and a run with
    densifications in it (the scene changes under the cut-offs)."""
    ref, s0, _ = _train(False, 24, scale, kind, densify=densify)
    got, s1, tr = _train(True, 24, scale, kind, densify=densify)
    for i, (a, b) in enumerate(zip(ref, got)):
        assert a.shape == b.shape and torch.equal(a, b), (i, float((a - b).abs().max()))
    assert s1.get("culled_frames", 0) - s0.get("culled_frames", 0) >= 12          # the cut-offs were applied
    if scale > 2:
        finite = sum(int((t != -1).sum()) for t in tr.tile_cull.values())
        assert finite > 0.5 * sum(t.numel() for t in tr.tile_cull.values())        # ... and most tiles of the dense scene have one


def test_a_truncation_that_is_too_tight_flags_the_frame_and_the_view_is_run_again():
    ref, _, _ = _train(False, 14, 2.2)
    got, stats, tr = _train(True, 14, 2.2, shrink_at=9)
    assert tr.rerun_views >= 1 and stats.get("cull_miss_frames", 0) >= 1
    for i, (a, b) in enumerate(zip(ref, got)):
        assert torch.equal(a, b), (i, float((a - b).abs().max()))


def test_truncated_frame_through_the_c_abi_is_the_untruncated_frame():
    """gsr_forward_async_culled directly: frame 1 (apply = 0) learns the cut-offs, frame 2 (apply = 1, unverified) renders with
    them: fewer instances, identical colour / inverse depth / radii / final_T / n_contrib, status word 6 clear; with cut-offs forced
    to zero the frame flags itself (word 6, and gsr_backward becomes a no-op: zero gradients)."""
    from diff_gaussian_rasterization import _C, GaussianRasterizationSettings, _settings_struct, _gauss_struct, _stream
    from scene_utils import make_gaussians, fibonacci_cameras
    from helpers import leaf_inputs, _view
    lib = _C.lib()
    dev = "cuda"
    raw = make_gaussians(4000, 1, seed=151, scale_factor=2.5)
    cam = fibonacci_cameras(2, 192, 128, seed=152)[0]
    W, H = 192, 128
    tiles = (W // 16) * (H // 16)
    inp = leaf_inputs(raw, torch.float32, dev, "sh")
    t = {k: v.detach().contiguous() for k, v in inp.items()}
    P = t["means3D"].shape[0]
    rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3, device=dev), 1.0,
                                       cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), 1,
                                       cam.camera_center.to(dev), False, False, False)
    s, keep = _settings_struct(rs, dev)
    g = _gauss_struct(P, t["means3D"], None, t["shs"], None, t["opacities"], t["scales"], t["rotations"], None)
    cap = 1 << 20
    cut = torch.full((tiles,), -1, dtype=torch.int32, device=dev)

    def frame(apply, cutoff):
        geom = torch.zeros(lib.gsr_geometry_state_bytes(P), dtype=torch.uint8, device=dev)
        img = torch.zeros(lib.gsr_image_state_bytes(W, H), dtype=torch.uint8, device=dev)
        binning = torch.zeros(lib.gsr_binning_state_bytes(P, W, H, cap), dtype=torch.uint8, device=dev)
        radii = torch.zeros(P, dtype=torch.int32, device=dev)
        color, invd = torch.empty(3, H, W, device=dev), torch.empty(1, H, W, device=dev)
        status = torch.zeros(4, dtype=torch.int64).pin_memory()
        _C.check(lib.gsr_forward_async_culled(C.byref(s), C.byref(g), _C.ptr(geom), geom.numel(), _C.ptr(radii), _C.ptr(binning),
                                              binning.numel(), cap, _C.ptr(img), img.numel(), _C.ptr(color), _C.ptr(invd), 1, 0,
                                              None, C.c_void_p(status.data_ptr()), 1, _stream(), None, _C.ptr(cutoff),
                                              1 if apply else 0))
        torch.cuda.synchronize()
        pi = [C.c_void_p() for _ in range(2)]
        lib.gsr_debug_image_views(_C.ptr(img), W, H, C.byref(pi[0]), C.byref(pi[1]))
        fT = _view(img, pi[0].value, W * H, torch.float32)
        nc = _view(img, pi[1].value, W * H, torch.int32)
        return dict(color=color.cpu(), invd=invd.cpu(), radii=radii.cpu(), fT=fT, nc=nc, R=int(status[1]),
                    miss=int(status[3]) & 0xFFFFFFFF, state=(geom, binning, img, radii))

    a = frame(False, cut)                       # untruncated; leaves the cut-offs
    learnt = cut.clone()
    assert int((learnt != -1).sum()) > tiles // 2
    b = frame(True, cut)                        # truncated by them
    assert b["miss"] == 0 and b["R"] < 0.8 * a["R"], (a["R"], b["R"])
    for k in ("color", "invd", "radii", "fT", "nc"):
        assert torch.equal(a[k], b[k]), k
    zero = torch.zeros_like(cut)
    c = frame(True, zero)                       # a cut-off in front of everything
    assert c["miss"] == 1 and c["R"] == 0
    # its backward is a no-op: zero gradients
    geom, binning, img, radii = c["state"]
    gr_t = {k: torch.full_like(v, 7.0) for k, v in (("m3", t["means3D"]), ("op", t["opacities"]), ("sc", t["scales"]),
                                                    ("ro", t["rotations"]), ("sh", t["shs"]))}
    m2 = torch.full((P, 3), 7.0, device=dev)
    gr = _C.gsr_grads(gr_t["m3"].data_ptr(), m2.data_ptr(), None, gr_t["sh"].data_ptr(), None, gr_t["op"].data_ptr(),
                      gr_t["sc"].data_ptr(), gr_t["ro"].data_ptr(), None, None, None, None)
    scratch = torch.zeros(lib.gsr_backward_scratch_bytes(P, cap), dtype=torch.uint8, device=dev)
    gcol = torch.ones(3, H, W, device=dev)
    _C.check(lib.gsr_backward(C.byref(s), C.byref(g), _C.ptr(radii), _C.ptr(geom), _C.ptr(binning), _C.ptr(img), cap, _C.ptr(gcol),
                              None, _C.ptr(scratch), scratch.numel(), C.byref(gr), _stream()))
    torch.cuda.synchronize()
    for k, v in gr_t.items():
        assert float(v.abs().max()) == 0.0, k
    assert float(m2.abs().max()) == 0.0
