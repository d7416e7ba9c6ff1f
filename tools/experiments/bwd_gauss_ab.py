#!/usr/bin/env python3
"""A/B of the per-Gaussian (bucketed) compositing backward (tools/experiments/bwd_gauss.hip) against the product's
k_render_bwd_tile on one view of BASELINE configs[2] (1 M Gaussians, SH3, 1920x1080): same forward state, same dL/dpixel.
Checks the challenger's per-Gaussian gradient sums against the product's gradients, then times both (HIP events).

    python tools/experiments/bwd_gauss_ab.py [--config 3] [--reps 10]           (GPU box, repo root)
"""
import argparse
import ctypes as C
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-slam_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

LOG2E = 1.4426950408889634


def dev_view(state, ptr, count, dtype):
    nbytes = count * torch.tensor([], dtype=dtype).element_size()
    off = ptr - state.data_ptr()
    assert 0 <= off and off + nbytes <= state.numel()
    return state[off:off + nbytes].view(dtype)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    from scene_utils import make_config
    from diff_gaussian_rasterization import _C, GaussianRasterizationSettings, _settings_struct, _gauss_struct, _stream
    lib = _C.lib()
    exp = C.CDLL(os.path.join(ROOT, "tools", "experiments", os.environ.get("BG_LIB", "libbwd_gauss.so")))
    dev = "cuda"
    raw, cams, cfg = make_config(args.config, views=2)
    cam = cams[1]
    P, W, H, deg = cfg["P"], cfg["W"], cfg["H"], cfg["deg"]
    act = raw.activated()
    t = {k: v.to(dev).float().contiguous() for k, v in act.items()}
    bg = torch.zeros(3, device=dev)
    rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg, 1.0,
                                       cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), deg,
                                       cam.camera_center.to(dev), False, False, False)
    s, keep = _settings_struct(rs, dev)
    g = _gauss_struct(P, t["means3D"], None, t["shs"], None, t["opacities"], t["scales"], t["rotations"], None)
    geom = torch.zeros(lib.gsr_geometry_state_bytes(P), dtype=torch.uint8, device=dev)
    img = torch.zeros(lib.gsr_image_state_bytes(W, H), dtype=torch.uint8, device=dev)
    radii = torch.zeros(P, dtype=torch.int32, device=dev)
    color, invd = torch.empty(3, H, W, device=dev), torch.empty(1, H, W, device=dev)
    R = _C.check(lib.gsr_forward_prepare(C.byref(s), C.byref(g), _C.ptr(geom), geom.numel(), _C.ptr(radii), _stream()))
    binning = torch.zeros(lib.gsr_binning_state_bytes(P, W, H, R), dtype=torch.uint8, device=dev)
    _C.check(lib.gsr_forward_render(C.byref(s), C.byref(g), _C.ptr(geom), _C.ptr(binning), binning.numel(), R, _C.ptr(img),
                                    img.numel(), _C.ptr(color), _C.ptr(invd), 1, _stream()))
    torch.cuda.synchronize()
    gx, gy = (W + 15) // 16, (H + 15) // 16
    tiles = gx * gy
    pv = [C.c_void_p() for _ in range(7)]
    lib.gsr_debug_geometry_views(_C.ptr(geom), P, *[C.byref(p) for p in pv])
    pb = [C.c_void_p() for _ in range(2)]
    lib.gsr_debug_binning_views(_C.ptr(binning), W, H, R, C.byref(pb[0]), C.byref(pb[1]))
    pi = [C.c_void_p() for _ in range(2)]
    lib.gsr_debug_image_views(_C.ptr(img), W, H, C.byref(pi[0]), C.byref(pi[1]))
    rec = dev_view(geom, pv[0].value, P * 12, torch.float32).view(P, 12)
    clamped = dev_view(geom, pv[6].value, P, torch.uint8)
    plist = dev_view(binning, pb[0].value, R, torch.int32)
    gen = torch.Generator().manual_seed(3)
    gpix = torch.randn(3, H, W, generator=gen).to(dev)

    # ---- product backward (gradients + timing of its compositing kernel through the library's own HIP-event profile) ----
    d_m3, d_m2 = torch.empty(P, 3, device=dev), torch.empty(P, 3, device=dev)
    d_op, d_sh = torch.empty(P, 1, device=dev), torch.empty_like(t["shs"])
    d_sc, d_ro = torch.empty(P, 3, device=dev), torch.empty(P, 4, device=dev)
    scratch = torch.empty(lib.gsr_backward_scratch_bytes(P, R), dtype=torch.uint8, device=dev)
    gr = _C.gsr_grads(*[None if x is None else x.data_ptr() for x in (d_m3, d_m2, None, d_sh, None, d_op, d_sc, d_ro, None, None,
                                                                        None, None)])
    lib.gsr_profile_enable(1)
    lib.gsr_profile_reset()
    for _ in range(args.reps):
        _C.check(lib.gsr_backward(C.byref(s), C.byref(g), _C.ptr(radii), _C.ptr(geom), _C.ptr(binning), _C.ptr(img), R,
                                  _C.ptr(gpix), None, _C.ptr(scratch), scratch.numel(), C.byref(gr), _stream()))
    torch.cuda.synchronize()
    prof = _C.profile_read()
    lib.gsr_profile_enable(0)
    product_ms = prof["render_bwd"][0] / prof["render_bwd"][1]

    # ---- challenger ----
    rows = R // 64 + tiles + 2
    ckpt = torch.empty(rows * 256, 2, device=dev)
    tile_todo = torch.zeros(tiles, dtype=torch.int32, device=dev)
    worklist = torch.zeros(rows, dtype=torch.int32, device=dev)
    work_count = torch.zeros(1, dtype=torch.int32, device=dev)
    igrad = torch.zeros(R, 12, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P_ = lambda x: C.c_void_p(x if isinstance(x, int) else x.data_ptr())       # noqa: E731

    def pre():
        rc = exp.bg_checkpoints(W, H, tiles, gx, P_(pb[1].value), P_(pb[0].value), P_(pv[0].value), P_(pi[1].value), P_(gpix),
                                P_(ckpt), P_(tile_todo), P_(worklist), P_(work_count), st)
        assert rc == 0, rc

    def bwd():
        rc = exp.bg_backward(W, H, gx, rows, P_(pb[1].value), P_(pb[0].value), P_(pv[0].value), P_(color), P_(pi[1].value),
                             P_(gpix), P_(ckpt), P_(tile_todo), P_(worklist), P_(work_count), P_(igrad), st)
        assert rc == 0, rc

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / args.reps

    pre_ms = timed(pre)
    bwd_ms = timed(bwd)
    nwork = int(work_count.item())

    # ---- correctness of the challenger: per-Gaussian sums against the product's gradients ----
    valid = plist >= 0
    ids = plist[valid].long()
    sums = torch.zeros(P, 12, device=dev, dtype=torch.float64).index_add_(0, ids, igrad[valid].double())
    cA, cB, cC = rec[:, 2].double() * (-2.0 / LOG2E), rec[:, 3].double() * (-1.0 / LOG2E), rec[:, 4].double() * (-2.0 / LOG2E)
    m2d_x = 0.5 * W * (-cA * sums[:, 0] - cB * sums[:, 1])
    m2d_y = 0.5 * H * (-cC * sums[:, 1] - cB * sums[:, 0])

    def rel(a, b):
        return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    checks = {"opacity": rel(sums[:, 5], d_op[:, 0]), "means2D_x": rel(m2d_x, d_m2[:, 0]), "means2D_y": rel(m2d_y, d_m2[:, 1])}
    C0 = 0.28209479177387814
    for c in range(3):
        keepc = ((clamped >> c) & 1) == 0
        checks[f"rgb{c}"] = rel(C0 * sums[keepc, 6 + c], d_sh[keepc, 0, c])
    out = {"workload": f"c{args.config}: P={P} {W}x{H}, view 1, R={R}", "product_render_bwd_ms": round(product_ms, 4),
           "challenger_bwd_ms": round(bwd_ms, 4), "challenger_checkpoint_prepass_ms": round(pre_ms, 4),
           "tile_buckets": nwork, "challenger_build": os.environ.get("BG_LIB", "libbwd_gauss.so"), "rel_l2_vs_product": {k: float(f"{v:.3e}") for k, v in checks.items()},
           "ratio": round(bwd_ms / product_ms, 2)}
    print(json.dumps(out))
    assert max(checks.values()) < 1e-3, checks


if __name__ == "__main__":
    main()
