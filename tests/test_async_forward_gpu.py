"""Speculative forward (gsr_forward_async: the whole frame enqueued for a capacity estimate, num_rendered on the device;
include/gsr.h) against the blocking pair gsr_forward_prepare + gsr_forward_render, which keeps the published rasterizer's
read-back in the middle of every frame (SURVEY.md 2.3 "D2H num_rendered"; reference call site
gaussian_renderer/__init__.py:90-109).

  "exact" (the DEFAULT mode): every frame must be the blocking path's frame bit for bit - also a frame with several times the
      instances its shape (P, W, H) has shown before (the count is verified before the call returns and phase 2 is repeated);
  "async": bit-identical whenever num_rendered <= capacity; beyond it the frame is composited from a truncated list, never writes
      out of bounds, its backward is a NO-OP (zero gradients; folded optimizer: parameters, moments, statistics keep their
      bits), it is reported per frame, and the Trainer runs the view again.
"""
import ctypes as C

import pytest
import torch

from helpers import run_hip, upstream_grads
from scene_utils import make_gaussians, fibonacci_cameras

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _restore_mode():
    import diff_gaussian_rasterization as dgr
    mode = dgr.forward_mode()
    yield
    dgr.set_forward_mode(mode)


def _views(n=5, P=6000, W=208, H=128):
    raw = make_gaussians(P, 3, seed=301, scale_factor=0.7)
    cams = fibonacci_cameras(n, W, H, seed=302)
    return raw, cams


def _thinned(raw, keep_every=5):
    """The same scene with all but every `keep_every`-th Gaussian made transparent (opacity far below 1/255: no tile
    instances): a frame of the same (P, W, H) with about 1 / keep_every of the instances."""
    import copy
    out = copy.deepcopy(raw)
    mask = torch.ones(out.opacity.shape[0], dtype=torch.bool)
    mask[::keep_every] = False
    out.opacity[mask] = -12.0
    return out


def _same(a, b):
    assert torch.equal(a["color"], b["color"]) and torch.equal(a["invdepth"], b["invdepth"])
    assert torch.equal(a["radii"], b["radii"])
    if a["grads"] is not None:
        for k in a["grads"]:
            assert torch.equal(a["grads"][k], b["grads"][k]), k


def test_default_mode_is_exact():
    import os
    import diff_gaussian_rasterization as dgr
    if "GSR_FORWARD_MODE" not in os.environ:
        assert dgr.forward_mode() == "exact"


@pytest.mark.parametrize("mode", ["exact", "async"])
def test_speculative_forward_and_backward_bit_identical_to_blocking(mode):
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _workspace as ws
    raw, cams = _views()
    bg = torch.tensor([0.3, 0.1, 0.2])
    gc, gd = upstream_grads(cams[0].image_height, cams[0].image_width)
    dgr.set_forward_mode("sync")
    ref = [run_hip(raw, c, 3, bg, gc=gc, gd=gd) for c in cams]
    dgr.set_forward_mode(mode)
    pool = ws.pool(torch.device("cuda", 0))
    pool.forget_estimates()                    # the first frame of the shape is speculative too (capacity guessed from P)
    before = dict(pool.stats)
    outs = [run_hip(raw, c, 3, bg, gc=gc, gd=gd) for c in cams]
    stats = dgr.call_stats()
    assert stats[mode + "_frames"] - before[mode + "_frames"] == len(cams)
    assert stats["sync_frames"] == before["sync_frames"]
    assert stats["overflow_frames"] == before["overflow_frames"]
    for a, b in zip(ref, outs):
        _same(a, b)
    # forward-only (torch.no_grad, reference render.py:49) takes the same route
    with torch.no_grad():
        c = run_hip(raw, cams[1], 3, bg)
    assert torch.equal(c["color"], ref[1]["color"])


@pytest.mark.parametrize("tile_local", [False, True])
def test_default_mode_frame_with_three_times_the_instances_is_exact(tile_local, monkeypatch):
    """DEFAULT mode: scene A (small splats) then scene B (the same P, W, H, more than 3x the tile instances), images and
    gradients of B and of a forward-only B against the blocking path, bit for bit, in both binning forms; B's phase 2 must
    have been repeated (its count exceeded what A had taught the pool)."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _workspace as ws
    monkeypatch.setattr(ws, "_BINNING", "tile" if tile_local else "global")
    P, W, H = 5000, 160, 96
    cam = fibonacci_cameras(2, W, H, seed=402)[0]
    bg = torch.tensor([0.05, 0.0, 0.1])
    gc, gd = upstream_grads(H, W)
    b = make_gaussians(P, 3, seed=403, scale_factor=0.9)
    a = _thinned(b)
    mode0 = dgr.forward_mode()
    dgr.set_forward_mode("sync")
    ref_a = run_hip(a, cam, 3, bg, gc=gc, gd=gd)
    Ra = dgr.call_stats()["num_rendered"]
    ref_b = run_hip(b, cam, 3, bg, gc=gc, gd=gd)
    Rb = dgr.call_stats()["num_rendered"]
    assert Rb >= 3 * Ra > 0, (Ra, Rb)
    dgr.set_forward_mode(mode0 if mode0 != "sync" else "exact")
    assert dgr.forward_mode() in ("exact",)            # what a caller gets without asking for anything
    pool = ws.pool(torch.device("cuda", 0))
    pool.forget_estimates()
    old_min, ws.MIN_CAPACITY = ws.MIN_CAPACITY, 256
    try:
        pool.capacity[(P, W, H)] = ws._capacity_for(Ra)    # as if earlier frames of the shape had looked like A
        for w in pool.free:                                  # fresh buffers of exactly that size: an overrun would corrupt
            w.binning = w.scratch = None
        out_a = run_hip(a, cam, 3, bg, gc=gc, gd=gd)
        n_re = dgr.call_stats()["rerendered_frames"]
        out_b = run_hip(b, cam, 3, bg, gc=gc, gd=gd)
        st = dgr.call_stats()
        assert st["rerendered_frames"] == n_re + 1 and st["num_rendered"] == Rb and st["overflow_frames"] == 0
        assert pool.capacity[(P, W, H)] >= Rb
        _same(ref_a, out_a)
        _same(ref_b, out_b)
        pool.capacity[(P, W, H)] = ws._capacity_for(Ra)
        with torch.no_grad():
            fo = run_hip(b, cam, 3, bg)                      # forward-only: waits for ITS count as well
        assert torch.equal(fo["color"], ref_b["color"]) and torch.equal(fo["invdepth"], ref_b["invdepth"])
    finally:
        ws.MIN_CAPACITY = old_min


def _trainer(raw, cams, gts, optimizer="hip_fused"):
    from scene_utils import GaussianModel, Trainer
    from gaussian_renderer import render, PipelineParams
    model = GaussianModel.from_raw(raw.to("cuda"), requires_grad=True)
    tr = Trainer(model, cams, gts, render, PipelineParams(), torch.zeros(3, device="cuda"), optimizer=optimizer, loss="hip",
                 separate_sh=True)
    return model, tr


def _model_state(model, tr):
    out = {n: p.detach().clone() for n, p in zip(("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"), model.parameters())}
    for n, p in zip(("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"), model.parameters()):
        st = tr.optimizer.state.get(p, {})
        if "exp_avg" in st:
            out["m_" + n], out["v_" + n] = st["exp_avg"].clone(), st["exp_avg_sq"].clone()
    out["accum"], out["denom"], out["maxr"] = model.xyz_gradient_accum.clone(), model.denom.clone(), model.max_radii2D.clone()
    return out


@pytest.mark.parametrize("tile_local", [False, True])
def test_training_step_after_an_instance_jump_default_and_async(tile_local, monkeypatch):
    """One training step (render + fused loss + backward with the optimizer folded in + statistics) on a frame with > 3x the
    instances the shape had shown: parameters, both moments and the densification statistics against the blocking mode, bit for
    bit - in the DEFAULT mode directly, and in "async" mode after the Trainer has run the truncated view again; the truncated
    async step itself must leave every one of those tensors bit-unchanged."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _workspace as ws
    from scene_utils import GaussianModel
    from gaussian_renderer import render, PipelineParams
    monkeypatch.setattr(ws, "_BINNING", "tile" if tile_local else "global")
    P, W, H = 5000, 160, 96
    cams = fibonacci_cameras(2, W, H, seed=412)
    for c in cams:
        c.to("cuda")
    big = make_gaussians(P, 3, seed=413, scale_factor=0.9)
    small = _thinned(big)
    bg = torch.zeros(3, device="cuda")
    pipe = PipelineParams()
    with torch.no_grad():
        teacher = GaussianModel.from_raw(make_gaussians(P, 3, seed=414, scale_factor=1.0).to("cuda"), requires_grad=False)
        gts = {v: render(cams[v], teacher, pipe, bg)["render"].clamp(0, 1).clone() for v in range(2)}
        small_model = GaussianModel.from_raw(small.to("cuda"), requires_grad=False)
    pool = ws.pool(torch.device("cuda", 0))
    key = (P, W, H)

    def taught_by_small_frames():
        """the pool as frames of the small-splat scene leave it"""
        pool.forget_estimates()
        for w in pool.free:
            w.binning = w.scratch = None
        with torch.no_grad():
            render(cams[0], small_model, pipe, bg)
        dgr.call_stats()
        return pool.capacity[key]

    old_min, ws.MIN_CAPACITY = ws.MIN_CAPACITY, 256
    try:
        # blocking reference: two steps
        dgr.set_forward_mode("sync")
        model, tr = _trainer(big, cams, gts)
        init = _model_state(model, tr)
        tr.step(0)
        ref1 = _model_state(model, tr)
        Rb = dgr.call_stats()["num_rendered"]
        tr.step(1)
        ref2 = _model_state(model, tr)

        # DEFAULT mode
        dgr.set_forward_mode("exact")
        cap = taught_by_small_frames()
        assert Rb >= 3 * cap / ws.HEADROOM, (Rb, cap)
        model, tr = _trainer(big, cams, gts)
        n_re = dgr.call_stats()["rerendered_frames"]
        tr.step(0)
        got = _model_state(model, tr)
        assert dgr.call_stats()["rerendered_frames"] == n_re + 1
        for k in ref1:
            assert torch.equal(got[k], ref1[k]), k
        tr.step(1)
        got = _model_state(model, tr)
        for k in ref2:
            assert torch.equal(got[k], ref2[k]), k

        # "async": the truncated step is a no-op, is reported, and the Trainer runs the view again
        dgr.set_forward_mode("async")
        taught_by_small_frames()
        model, tr = _trainer(big, cams, gts)
        with pytest.warns(RuntimeWarning, match="truncated"):
            tr.step(0)
            torch.cuda.synchronize()
            unchanged = _model_state(model, tr)
            for k, v in init.items():
                assert torch.equal(unchanged[k], v), k
            for k in unchanged:
                if k[:2] in ("m_", "v_"):
                    assert not unchanged[k].any(), k
            assert dgr.call_stats()["overflow_frames"] >= 1
            tr.finish()                                       # -> _rerun_truncated_frames(wait=True)
        assert tr.rerun_views == 1
        got = _model_state(model, tr)
        for k in ref1:
            assert torch.equal(got[k], ref1[k]), k
        tr.step(1)
        tr.finish()
        got = _model_state(model, tr)
        for k in ref2:
            assert torch.equal(got[k], ref2[k]), k
    finally:
        ws.MIN_CAPACITY = old_min


@pytest.mark.parametrize("tile_local", [False, True])
def test_async_capacity_overflow_is_contained_reported_and_heals(tile_local, monkeypatch):
    """"async" with a capacity far below the instance count: the frame renders from a truncated list (nearest-first in the
    global binning form, lowest ids first in the tile-local form) without touching memory outside its buffers, every gradient
    of its backward is an exact zero, the overflow is counted and reported by ticket, the capacity is raised, and the
    following frame is exact."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _workspace as ws
    monkeypatch.setattr(ws, "_BINNING", "tile" if tile_local else "global")
    raw, cams = _views(n=2, P=5000, W=160, H=96)
    bg = torch.zeros(3)
    gc, gd = upstream_grads(96, 160)
    dgr.set_forward_mode("sync")
    ref = run_hip(raw, cams[0], 3, bg, gc=gc, gd=gd)
    R = dgr.call_stats()["num_rendered"]
    assert R > 4096
    dgr.set_forward_mode("async")
    pool = ws.pool(torch.device("cuda", 0))
    key = (5000, 160, 96)
    old_min, ws.MIN_CAPACITY = ws.MIN_CAPACITY, 256
    try:
        pool.capacity[key] = max(256, R // 3)            # a third of what the view needs
        for w in pool.free:                                # fresh buffers of exactly that size: an overrun would fault / corrupt
            w.binning = w.scratch = None
        n0 = dgr.call_stats()["overflow_frames"]
        dgr.take_overflowed()
        with pytest.warns(RuntimeWarning, match="truncated"):
            out = run_hip(raw, cams[0], 3, bg, gc=gc, gd=gd)
            ticket = dgr.last_ticket()
            st = dgr.call_stats()
        assert st["overflow_frames"] == n0 + 1 and st["num_rendered"] == R
        assert dgr.take_overflowed() == [ticket] and dgr.take_overflowed() == []
        assert pool.capacity[key] >= R
        assert torch.isfinite(out["color"]).all() and float(out["color"].min()) >= 0.0
        assert torch.equal(out["radii"], ref["radii"])     # per-Gaussian outputs do not depend on the binning capacity
        for k, g in out["grads"].items():                  # nothing is learnt from a truncated frame
            assert not g.any(), k
        again = run_hip(raw, cams[0], 3, bg, gc=gc, gd=gd)
        assert dgr.call_stats()["overflow_frames"] == n0 + 1
        _same(ref, again)
    finally:
        ws.MIN_CAPACITY = old_min


def _lowlevel_setup(P=4000, W=176, H=112):
    from diff_gaussian_rasterization import _C, GaussianRasterizationSettings, _settings_struct, _gauss_struct
    from helpers import settings_for, leaf_inputs, lowlevel_forward
    raw, cams = _views(n=1, P=P, W=W, H=H)
    cam, bg = cams[0], torch.tensor([0.1, 0.1, 0.1])
    a = lowlevel_forward(raw, cam, 3, bg)
    lib = _C.lib()
    dev = "cuda"
    inp = leaf_inputs(raw, torch.float32, dev, "sh")
    rs = settings_for(cam, 3, bg, 1.0, False, cls=GaussianRasterizationSettings, device=dev)
    s, keep = _settings_struct(rs, dev)
    t = {k: v.detach().contiguous() for k, v in inp.items()}
    g = _gauss_struct(P, t["means3D"], None, t["shs"], None, t["opacities"], t["scales"], t["rotations"], None)
    return a, lib, s, g, (keep, t, inp, rs)


def _check_state(lib, a, binning, cap, W, H, color, invd, radii):
    from diff_gaussian_rasterization import _C
    from helpers import _view
    assert torch.equal(color.cpu(), a["color"]) and torch.equal(invd.cpu(), a["invdepth"]) and torch.equal(radii.cpu(), a["radii"])
    pb = [C.c_void_p() for _ in range(2)]
    lib.gsr_debug_binning_views(_C.ptr(binning), W, H, cap, C.byref(pb[0]), C.byref(pb[1]))
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    pl = _view(binning, pb[0].value, a["R"], torch.int32).numpy().view("uint32")
    rg = _view(binning, pb[1].value, tiles * 2, torch.int32).numpy().view("uint32").reshape(tiles, 2)
    assert (pl == a["point_list"]).all() and (rg == a["ranges"]).all()
    return rg


@pytest.mark.parametrize("tlo,pinned,verify", [(0, True, False), (1, True, False), (1, False, False), (0, True, True),
                                               (1, True, True)])
def test_lowlevel_async_call_matches_blocking_state(tlo, pinned, verify):
    """The C ABI directly: gsr_forward_async with a generous capacity - in both binning forms (tlo = 1: emission in index
    order + per-tile depth ordering in LDS) - leaves the same images, the same sorted lists (first num_rendered entries) and the
    same tile ranges as the blocking calls.  The status words reach a pinned slot through the compositing kernel's own store
    and pageable memory through a copy; with `num_rendered_out` the call itself returns the count."""
    from diff_gaussian_rasterization import _C, _stream
    P, W, H = 4000, 176, 112
    a, lib, s, g, keep = _lowlevel_setup(P, W, H)
    dev = "cuda"
    cap = int(a["R"] * 1.5) + 1000
    geom = torch.zeros(lib.gsr_geometry_state_bytes(P), dtype=torch.uint8, device=dev)
    img = torch.zeros(lib.gsr_image_state_bytes(W, H), dtype=torch.uint8, device=dev)
    binning = torch.zeros(lib.gsr_binning_state_bytes(P, W, H, cap), dtype=torch.uint8, device=dev)
    radii = torch.zeros(P, dtype=torch.int32, device=dev)
    color, invd = torch.empty(3, H, W, device=dev), torch.empty(1, H, W, device=dev)
    status = torch.zeros(4, dtype=torch.int64)
    if pinned:
        status = status.pin_memory()
    count = C.c_int64(-1)
    _C.check(lib.gsr_forward_async(C.byref(s), C.byref(g), _C.ptr(geom), geom.numel(), _C.ptr(radii), _C.ptr(binning),
                                   binning.numel(), cap, _C.ptr(img), img.numel(), _C.ptr(color), _C.ptr(invd), 1, 0, None,
                                   C.c_void_p(status.data_ptr()), tlo, _stream(), C.byref(count) if verify else None))
    if verify:
        assert count.value == a["R"]                     # known before the device has finished the frame
    torch.cuda.synchronize()
    assert int(status[1]) == a["R"]
    rg = _check_state(lib, a, binning, cap, W, H, color, invd, radii)
    if tlo:      # the longest tile list is reported once it passes half the LDS capacity of the per-tile sort (else 0)
        longest = int((rg[:, 1] - rg[:, 0]).max())
        assert int(status[2]) & 0xFFFFFFFF == (longest if longest > 2048 else 0)


@pytest.mark.parametrize("tlo", [0, 1])
def test_lowlevel_verified_overflow_then_rerender_matches_blocking_state(tlo):
    """gsr_forward_async(num_rendered_out) with a third of the needed capacity reports the true count; gsr_forward_rerender on a
    binning state that holds it then leaves exactly the blocking path's images, lists and ranges."""
    from diff_gaussian_rasterization import _C, _stream
    P, W, H = 4000, 176, 112
    a, lib, s, g, keep = _lowlevel_setup(P, W, H)
    dev = "cuda"
    small = max(256, a["R"] // 3)
    geom = torch.zeros(lib.gsr_geometry_state_bytes(P), dtype=torch.uint8, device=dev)
    img = torch.zeros(lib.gsr_image_state_bytes(W, H), dtype=torch.uint8, device=dev)
    binning = torch.zeros(lib.gsr_binning_state_bytes(P, W, H, small), dtype=torch.uint8, device=dev)
    radii = torch.zeros(P, dtype=torch.int32, device=dev)
    color, invd = torch.empty(3, H, W, device=dev), torch.empty(1, H, W, device=dev)
    status = torch.zeros(4, dtype=torch.int64).pin_memory()
    count = C.c_int64(-1)
    _C.check(lib.gsr_forward_async(C.byref(s), C.byref(g), _C.ptr(geom), geom.numel(), _C.ptr(radii), _C.ptr(binning),
                                   binning.numel(), small, _C.ptr(img), img.numel(), _C.ptr(color), _C.ptr(invd), 1, 0, None,
                                   C.c_void_p(status.data_ptr()), tlo, _stream(), C.byref(count)))
    assert count.value == a["R"] > small
    cap = a["R"] + 17
    big = torch.zeros(lib.gsr_binning_state_bytes(P, W, H, cap), dtype=torch.uint8, device=dev)
    _C.check(lib.gsr_forward_rerender(C.byref(s), C.byref(g), _C.ptr(geom), _C.ptr(big), big.numel(), cap, _C.ptr(img),
                                      img.numel(), _C.ptr(color), _C.ptr(invd), 1, tlo, C.c_void_p(status.data_ptr()),
                                      _stream()))
    torch.cuda.synchronize()
    assert int(status[1]) == a["R"]
    _check_state(lib, a, big, cap, W, H, color, invd, radii)


@pytest.mark.parametrize("tlo", [0, 1])
def test_lowlevel_forward_ignores_state_contents_and_reports_prefiltered_culls(tlo):
    """The three state buffers are opaque scratch: a forward into buffers full of 0xFF leaves the blocking path's results (the
    tile-local form runs no memset at all: its scan kernel writes every status word).  And `prefiltered` with a point behind the
    near plane is the published hard error in both forms - returned by the verified call, flagged in the status words of the
    unverified one (tile-local form: the flag travels in the projection workgroups' instance totals)."""
    from diff_gaussian_rasterization import _C, _stream, _settings_struct, _gauss_struct
    P, W, H = 4000, 176, 112
    a, lib, s, g, keep = _lowlevel_setup(P, W, H)
    dev = "cuda"
    cap = int(a["R"] * 1.5) + 1000

    def buffers():
        geom = torch.full((lib.gsr_geometry_state_bytes(P),), 0xFF, dtype=torch.uint8, device=dev)
        img = torch.full((lib.gsr_image_state_bytes(W, H),), 0xFF, dtype=torch.uint8, device=dev)
        binning = torch.full((lib.gsr_binning_state_bytes(P, W, H, cap),), 0xFF, dtype=torch.uint8, device=dev)
        radii = torch.full((P,), -1, dtype=torch.int32, device=dev)
        return geom, img, binning, radii, torch.empty(3, H, W, device=dev), torch.empty(1, H, W, device=dev)

    def call(s_, g_, bufs, verify):
        geom, img, binning, radii, color, invd = bufs
        status = torch.full((4,), -1, dtype=torch.int64).pin_memory()
        count = C.c_int64(-1)
        rc = lib.gsr_forward_async(C.byref(s_), C.byref(g_), _C.ptr(geom), geom.numel(), _C.ptr(radii), _C.ptr(binning),
                                   binning.numel(), cap, _C.ptr(img), img.numel(), _C.ptr(color), _C.ptr(invd), 1, 0, None,
                                   C.c_void_p(status.data_ptr()), tlo, _stream(), C.byref(count) if verify else None)
        torch.cuda.synchronize()
        return rc, count.value, status

    for verify in (True, False):
        bufs = buffers()
        rc, n, status = call(s, g, bufs, verify)
        assert rc == 0 and int(status[1]) == a["R"] and (not verify or n == a["R"])
        assert (int(status[0]) >> 32) & 1 == 0                      # no cull flag
        _check_state(lib, a, bufs[2], cap, W, H, bufs[4], bufs[5], bufs[3])
    # one Gaussian moved into the camera centre: behind the near plane
    _, t, inp, rs = keep
    means = t["means3D"].clone()
    means[P // 2] = rs.campos.to(means)
    s2, keep2 = _settings_struct(rs._replace(prefiltered=True), dev)
    g2 = _gauss_struct(P, means, None, t["shs"], None, t["opacities"], t["scales"], t["rotations"], None)
    rc, n, status = call(s2, g2, buffers(), True)
    assert rc != 0 and "filtered" in _C.last_error()
    rc, n, status = call(s2, g2, buffers(), False)
    assert rc == 0 and (int(status[0]) >> 32) & 1 == 1
    # and the same scene without `prefiltered` is simply rendered
    s3, keep3 = _settings_struct(rs, dev)
    rc, n, status = call(s3, g2, buffers(), True)
    assert rc == 0 and (int(status[0]) >> 32) & 1 == 0 and n > 0


def test_default_path_files_the_walk_classes_of_every_frame():
    """Round 4: in the default (tile-local, non-blocking) forward the walk-class counters are cleared by the first workgroup of
    the per-tile ordering kernel, not by a memset.  If that clearing were lost the backward would find class sizes that do not add
    up to the grid and fall back to index order - same bits, no test would fail, only the speed would be gone.  So look at the
    state the Python path really used: after several training-mode renders of different views through one workspace pool, every
    workspace's counters add up to exactly one frame's tiles."""
    import ctypes as C
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _C, _workspace as ws
    if ws._BINNING != "tile":
        pytest.skip("GSR_BINNING=global")
    raw = make_gaussians(4000, 1, seed=901, scale_factor=1.0)
    cams = fibonacci_cameras(5, 208, 144, seed=902)
    bg = torch.tensor([0.1, 0.1, 0.1])
    gc, gd = upstream_grads(144, 208, depth=False)
    for cam in cams:                                             # (forward + backward each: the pool recycles its workspaces)
        run_hip(raw, cam, 1, bg, gc=gc, gd=None)
    torch.cuda.synchronize()
    pool = ws.pool(torch.device("cuda", torch.cuda.current_device()))
    tiles = ((208 + 15) // 16) * ((144 + 15) // 16)
    seen = 0
    for w in pool.free:
        if w.img is None or w.img.numel() < _C.lib().gsr_image_state_bytes(208, 144):
            continue
        pw = [C.c_void_p() for _ in range(3)]
        classes = _C.lib().gsr_debug_walk_views(_C.ptr(w.img), 208, 144, C.byref(pw[0]), C.byref(pw[1]), C.byref(pw[2]))
        off = pw[0].value - w.img.data_ptr()
        cnt = w.img[off:off + 4 * classes].view(torch.int32).cpu()
        assert int(cnt.sum()) == tiles, (cnt.tolist(), tiles)
        seen += 1
    assert seen >= 1


def test_parity_suite_with_the_other_backward_form_and_other_forward_modes():
    """The two forms of the compositing backward (one wave per tile from 6000 tiles up, four waves per tile below) are chosen by
    image size, so the small-image parity tests only ever see the four-wave form.  Re-run the core parity tests in a child
    process with GSR_BWD_FORM=tile (a second child takes its opt-in matrix-pipe reduction, GSR_BWD_REDUCE=mfma:
    k_render_bwd_tile_mx) and the unverified forward (GSR_FORWARD_MODE=async; two parity modes, edge cases,
    bitwise repeat, committed golden), and once more with the blocking forward (GSR_FORWARD_MODE=sync), the global binning form and the
    colour pass on the side stream."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (two of the six parity modes per child - SH with anti-aliasing, precomputed colours with anti-aliasing and cov3D_precomp -
    # keep the children at ~30 s each: the float64 oracle is what takes the time)
    par = "tests/test_parity_gpu.py::test_forward_backward_parity"
    sel = [par + "[sh-True-False]", par + "[colors-True-True]", "tests/test_parity_gpu.py::test_edge_cases",
           "tests/test_parity_gpu.py::test_bitwise_reproducible", "tests/test_parity_gpu.py::test_against_committed_golden"]
    for extra, tests in (({"GSR_BWD_FORM": "tile", "GSR_FORWARD_MODE": "async"}, sel),
                         ({"GSR_BWD_FORM": "tile", "GSR_BWD_REDUCE": "mfma"}, sel[0:3]),     # (the opt-in matrix-pipe form of that kernel)
                         ({"GSR_BWD_FORM": "quad", "GSR_FORWARD_MODE": "sync", "GSR_SHADE_STREAM": "1"}, sel[1:4]),
                         ({"GSR_BINNING": "global", "GSR_SHADE_STREAM": "1"}, sel[2:4])):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu"] + tests, cwd=root, env=env,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (extra, r.stdout[-3000:], r.stderr[-2000:])


@pytest.mark.parametrize("P", [9000, 2600])
def test_tile_local_sort_long_lists_and_policy(P):
    """Every Gaussian piled onto the image centre.  P = 9000: tile lists of > 4096 entries take k_tile_depth_sort's in-memory
    path, and the Python policy (tile-local form first, back to the global form once a shape has shown lists beyond 3072
    entries) switches.  P = 2600: lists between 1024 and 3072 entries - the second launch of the LDS sort - and no switch.
    Either way every frame must give the blocking path's images and gradients bit for bit."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _workspace as ws
    if ws._BINNING != "tile":
        pytest.skip("GSR_BINNING=global")
    raw = make_gaussians(P, 1, seed=511, scale_factor=0.5)
    raw.xyz *= 0.02                                   # a 5 cm blob at the origin: every splat covers the central tiles
    cam = fibonacci_cameras(2, 96, 64, seed=512)[0]
    bg = torch.tensor([0.0, 0.1, 0.0])
    gc, gd = upstream_grads(64, 96)
    mode = dgr.forward_mode()
    dgr.set_forward_mode("sync")
    ref = run_hip(raw, cam, 1, bg, gc=gc, gd=gd)
    dgr.set_forward_mode("exact" if mode == "sync" else mode)
    pool = ws.pool(torch.device("cuda", 0))
    key = (P, 96, 64)
    pool.forget_estimates()
    n0 = pool.stats.get("tile_local_frames", 0)
    for it in range(4):
        out = run_hip(raw, cam, 1, bg, gc=gc, gd=gd)
        assert torch.equal(out["color"], ref["color"]) and torch.equal(out["invdepth"], ref["invdepth"]), it
        for k in ref["grads"]:
            assert torch.equal(out["grads"][k], ref["grads"][k]), (it, k)
        dgr.call_stats()                              # waits for the frame's status: the policy sees the long lists
    used = pool.stats.get("tile_local_frames", 0) - n0
    if P == 9000:
        assert pool.longest_list[key] > 4096
        assert 1 <= used < 4                          # first frame(s) tile-local (slow path inside), then the global form
    else:
        assert 2048 < pool.longest_list[key] <= ws.TLO_MAX_LIST and used >= 3
