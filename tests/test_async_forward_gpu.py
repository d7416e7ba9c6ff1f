"""Non-blocking forward (gsr_forward_async: device-side num_rendered, capacity-sized binning state, include/gsr.h) against the
blocking pair gsr_forward_prepare + gsr_forward_render, which keeps the published rasterizer's one read-back per frame.
Whenever num_rendered <= capacity the two must agree bit for bit (same kernels, same order); beyond the capacity the frame
loses its farthest instances, never writes out of bounds, and the next frame is exact again."""
import pytest
import torch

from helpers import run_hip, upstream_grads
from scene_utils import make_gaussians, fibonacci_cameras

pytestmark = pytest.mark.gpu


def _views(n=5, P=6000, W=208, H=128):
    raw = make_gaussians(P, 3, seed=301, scale_factor=0.7)
    cams = fibonacci_cameras(n, W, H, seed=302)
    return raw, cams


def test_async_forward_and_backward_bit_identical_to_blocking():
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _workspace as ws
    raw, cams = _views()
    bg = torch.tensor([0.3, 0.1, 0.2])
    gc, gd = upstream_grads(cams[0].image_height, cams[0].image_width)
    dgr.set_forward_mode("sync")
    try:
        ref = [run_hip(raw, c, 3, bg, gc=gc, gd=gd) for c in cams]
        dgr.set_forward_mode("async")
        pool = ws.pool(torch.device("cuda", 0))
        before = dict(pool.stats)
        outs = [run_hip(raw, c, 3, bg, gc=gc, gd=gd) for c in cams]       # first call of the shape may block, the rest do not
        stats = dgr.call_stats()
        assert stats["async_frames"] - before["async_frames"] >= len(cams) - 1
        assert stats["overflow_frames"] == before["overflow_frames"]
        for a, b in zip(ref, outs):
            assert torch.equal(a["color"], b["color"]) and torch.equal(a["invdepth"], b["invdepth"])
            assert torch.equal(a["radii"], b["radii"])
            for k in a["grads"]:
                assert torch.equal(a["grads"][k], b["grads"][k]), k
        # forward-only (torch.no_grad) takes the same route
        with torch.no_grad():
            c = run_hip(raw, cams[1], 3, bg)
        assert torch.equal(c["color"], ref[1]["color"])
    finally:
        dgr.set_forward_mode("async")


@pytest.mark.parametrize("tile_local", [False, True])
def test_async_capacity_overflow_is_contained_and_heals(tile_local, monkeypatch):
    """Force a capacity far below the instance count: the frame renders from a truncated list (nearest-first in the global
    binning form, lowest ids first in the tile-local form) without touching memory outside its buffers, the overflow is
    counted, the capacity is raised, and the following frame is exact."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _workspace as ws
    monkeypatch.setattr(ws, "TLO_SETTLE_FRAMES", 0 if tile_local else 1 << 30)
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
        out = run_hip(raw, cams[0], 3, bg, gc=gc, gd=gd)
        st = dgr.call_stats()
        assert st["overflow_frames"] == n0 + 1 and st["num_rendered"] == R
        assert pool.capacity[key] >= R
        assert torch.isfinite(out["color"]).all() and all(torch.isfinite(g).all() for g in out["grads"].values())
        assert torch.equal(out["radii"], ref["radii"])     # per-Gaussian outputs do not depend on the binning capacity
        # transmittance can only be higher with instances missing: a black background never gets brighter than the reference
        # by more than rounding... (colour is not monotone per channel, so only sanity-check the range)
        assert float(out["color"].min()) >= 0.0
        again = run_hip(raw, cams[0], 3, bg, gc=gc, gd=gd)
        assert dgr.call_stats()["overflow_frames"] == n0 + 1
        assert torch.equal(again["color"], ref["color"])
        for k in ref["grads"]:
            assert torch.equal(again["grads"][k], ref["grads"][k]), k
    finally:
        ws.MIN_CAPACITY = old_min


@pytest.mark.parametrize("tlo,pinned", [(0, True), (1, True), (1, False)])
def test_lowlevel_async_call_matches_blocking_state(tlo, pinned):
    """The C ABI directly: gsr_forward_async with a generous capacity - in both binning forms (tlo = 1: emission in index
    order + per-tile depth ordering in LDS) - leaves the same images, the same sorted lists (first num_rendered entries) and the
    same tile ranges as the blocking calls.  The status words reach a pinned slot through the compositing kernel's own store
    and pageable memory through a copy."""
    import ctypes as C
    import math
    from diff_gaussian_rasterization import _C, GaussianRasterizationSettings, _settings_struct, _gauss_struct, _stream
    from helpers import settings_for, leaf_inputs, lowlevel_forward, _view
    raw, cams = _views(n=1, P=4000, W=176, H=112)
    cam, bg = cams[0], torch.tensor([0.1, 0.1, 0.1])
    a = lowlevel_forward(raw, cam, 3, bg)
    lib = _C.lib()
    dev = "cuda"
    inp = leaf_inputs(raw, torch.float32, dev, "sh")
    P, H, W = 4000, cam.image_height, cam.image_width
    rs = settings_for(cam, 3, bg, 1.0, False, cls=GaussianRasterizationSettings, device=dev)
    s, keep = _settings_struct(rs, dev)
    t = {k: v.detach().contiguous() for k, v in inp.items()}
    g = _gauss_struct(P, t["means3D"], None, t["shs"], None, t["opacities"], t["scales"], t["rotations"], None)
    cap = int(a["R"] * 1.5) + 1000
    geom = torch.zeros(lib.gsr_geometry_state_bytes(P), dtype=torch.uint8, device=dev)
    img = torch.zeros(lib.gsr_image_state_bytes(W, H), dtype=torch.uint8, device=dev)
    binning = torch.zeros(lib.gsr_binning_state_bytes(P, W, H, cap), dtype=torch.uint8, device=dev)
    radii = torch.zeros(P, dtype=torch.int32, device=dev)
    color, invd = torch.empty(3, H, W, device=dev), torch.empty(1, H, W, device=dev)
    status = torch.zeros(4, dtype=torch.int64)
    if pinned:
        status = status.pin_memory()
    _C.check(lib.gsr_forward_async(C.byref(s), C.byref(g), _C.ptr(geom), geom.numel(), _C.ptr(radii), _C.ptr(binning),
                                   binning.numel(), cap, _C.ptr(img), img.numel(), _C.ptr(color), _C.ptr(invd), 1, 0, None,
                                   C.c_void_p(status.data_ptr()), tlo, _stream()))
    torch.cuda.synchronize()
    assert int(status[1]) == a["R"]
    assert torch.equal(color.cpu(), a["color"]) and torch.equal(invd.cpu(), a["invdepth"]) and torch.equal(radii.cpu(), a["radii"])
    pb = [C.c_void_p() for _ in range(2)]
    lib.gsr_debug_binning_views(_C.ptr(binning), W, H, cap, C.byref(pb[0]), C.byref(pb[1]))
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    pl = _view(binning, pb[0].value, a["R"], torch.int32).numpy().view("uint32")
    rg = _view(binning, pb[1].value, tiles * 2, torch.int32).numpy().view("uint32").reshape(tiles, 2)
    assert (pl == a["point_list"]).all() and (rg == a["ranges"]).all()
    if tlo:      # the longest tile list is reported once it passes half the LDS capacity of the per-tile sort (else 0)
        longest = int((rg[:, 1] - rg[:, 0]).max())
        assert int(status[2]) & 0xFFFFFFFF == (longest if longest > 2048 else 0)


def test_parity_suite_with_the_other_backward_form_and_blocking_forward():
    """The two forms of the compositing backward (one wave per tile from 6000 tiles up, four waves per tile below) are chosen by
    image size, so the small-image parity tests only ever see the four-wave form.  Re-run the core parity tests in a child
    process with GSR_BWD_FORM=tile (two parity modes, edge cases, bitwise repeat, committed golden), and once
    more with the blocking forward (GSR_FORWARD_MODE=sync) and the colour pass kept on the caller's stream."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (two of the six parity modes per child - SH with anti-aliasing, precomputed colours with anti-aliasing and cov3D_precomp -
    # keep the children at ~30 s each: the float64 oracle is what takes the time)
    par = "tests/test_parity_gpu.py::test_forward_backward_parity"
    sel = [par + "[sh-True-False]", par + "[colors-True-True]", "tests/test_parity_gpu.py::test_edge_cases",
           "tests/test_parity_gpu.py::test_bitwise_reproducible", "tests/test_parity_gpu.py::test_against_committed_golden"]
    for extra, tests in (({"GSR_BWD_FORM": "tile", "GSR_TLO_SETTLE": "0"}, sel),      # + tile-local binning from frame 2 on
                         ({"GSR_BWD_FORM": "quad", "GSR_FORWARD_MODE": "sync", "GSR_SHADE_STREAM": "1"}, sel[1:4])):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu"] + tests, cwd=root, env=env,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (extra, r.stdout[-3000:], r.stderr[-2000:])


@pytest.mark.parametrize("P", [9000, 2600])
def test_tile_local_sort_long_lists_and_policy(P, monkeypatch):
    """Every Gaussian piled onto the image centre.  P = 9000: tile lists of > 4096 entries take k_tile_depth_sort's in-memory
    path, and the Python policy (tile-local form first, back to the global form once a shape has shown lists beyond 3072
    entries) switches.  P = 2600: lists between 1024 and 3072 entries - the second launch of the LDS sort - and no switch.
    Either way every frame must give the blocking path's images and gradients bit for bit."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _workspace as ws
    if ws._BINNING != "tile":
        pytest.skip("GSR_BINNING=global")
    monkeypatch.setattr(ws, "TLO_SETTLE_FRAMES", 0)   # (normally the form waits until the shape's capacity has held 3 frames)
    raw = make_gaussians(P, 1, seed=511, scale_factor=0.5)
    raw.xyz *= 0.02                                   # a 5 cm blob at the origin: every splat covers the central tiles
    cam = fibonacci_cameras(2, 96, 64, seed=512)[0]
    bg = torch.tensor([0.0, 0.1, 0.0])
    gc, gd = upstream_grads(64, 96)
    dgr.set_forward_mode("sync")
    ref = run_hip(raw, cam, 1, bg, gc=gc, gd=gd)
    dgr.set_forward_mode("async")
    pool = ws.pool(torch.device("cuda", 0))
    key = (P, 96, 64)
    pool.longest_list.pop(key, None)
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
