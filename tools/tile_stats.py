#!/usr/bin/env python3
"""Distribution of the per-tile work of the compositing backward for one view of a BASELINE config (GPU box):
list length per tile, entries the backward actually walks (toDo = deepest contributor of any pixel of the tile), and the share of
the frame's work that sits in the longest tiles - what bounds a one-wave-per-tile launch from below.

    python tools/tile_stats.py [--config 3] [--scale-factor 1.0] [--view 0]
"""
import argparse
import ctypes as C
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-slam_amd")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--scale-factor", type=float, default=1.0)
    ap.add_argument("--view", type=int, default=0)
    a = ap.parse_args()
    from scene_utils import make_config
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _C, _settings_struct, _gauss_struct, _stream, GaussianRasterizationSettings
    lib = _C.lib()
    raw, cams, c = make_config(a.config, splat_scale=a.scale_factor)
    dev = "cuda"
    cam = cams[a.view]
    act = raw.activated()
    t = {k: v.to(dev).float().contiguous() for k, v in act.items()}
    P, W, H = t["means3D"].shape[0], c["W"], c["H"]
    rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3, device=dev), 1.0,
                                       cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), c["deg"],
                                       cam.camera_center.to(dev), False, False, bool(c.get("antialiasing", False)))
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
    pb = [C.c_void_p() for _ in range(2)]
    lib.gsr_debug_binning_views(_C.ptr(binning), W, H, R, C.byref(pb[0]), C.byref(pb[1]))
    off = pb[1].value - binning.data_ptr()
    ranges = binning[off:off + tiles * 8].view(torch.int32).view(tiles, 2).cpu().numpy().astype(np.int64)
    pi = [C.c_void_p() for _ in range(2)]
    lib.gsr_debug_image_views(_C.ptr(img), W, H, C.byref(pi[0]), C.byref(pi[1]))
    off = pi[1].value - img.data_ptr()
    ncon = img[off:off + 4 * W * H].view(torch.int32).view(H, W).cpu().numpy()
    length = ranges[:, 1] - ranges[:, 0]
    pad = np.zeros((gy * 16, gx * 16), dtype=np.int64)
    pad[:H, :W] = ncon
    per_tile = pad.reshape(gy, 16, gx, 16).transpose(0, 2, 1, 3).reshape(tiles, 256)
    todo = np.minimum(per_tile.max(axis=1), length)
    sub = pad.reshape(gy, 2, 8, gx, 2, 8).transpose(0, 3, 1, 4, 2, 5).reshape(tiles, 4, 64).max(axis=2)      # deepest contributor per 8x8 sub-block
    work = np.minimum(sub, length[:, None]).sum(axis=1)            # (entry, live sub-block) pairs the backward may have to look at

    def q(x):
        x = np.sort(x)
        return {"mean": float(x.mean()), "p50": int(x[len(x) // 2]), "p90": int(x[int(len(x) * 0.9)]), "p99": int(x[int(len(x) * 0.99)]),
                "max": int(x[-1])}
    out = {"config": a.config, "scale_factor": a.scale_factor, "P": P, "W": W, "H": H, "tiles": tiles, "num_rendered": int(R),
           "list_length": q(length), "entries_walked_by_the_backward": q(todo), "entry_x_live_subblock": q(work),
           "sum_entries_walked": int(todo.sum()), "pixel_entry_pairs": int(ncon.sum()),
           "longest_tile_over_mean_walk": float(todo.max() / max(1.0, todo.mean())),
           "one_wave_per_tile_bound": "duration >= walk_max x (time a lone wave needs per entry)"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
