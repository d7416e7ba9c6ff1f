"""Generates tests/golden/oracle_small_scene.npz from the float64 CPU oracle (oracle/gs_oracle.py).

The reference holds no golden vectors for the rasterizer (parity unpinned), so these are the
build's own: inputs (activated Gaussians, camera matrices, upstream gradients) and expected outputs (image,
inverse depth, radii, all gradients) for a 400-Gaussian, 80x56 (ragged tiles), SH-degree-3 scene with
anti-aliasing off and on.  CPU tests check the oracle still reproduces them; GPU tests check the HIP path
against them without running the oracle.      python tests/golden/make_oracle_fixtures.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-slam_amd"), os.path.dirname(HERE)):
    sys.path.insert(0, p)

from helpers import run_oracle, upstream_grads  # noqa: E402
from scene_utils import make_gaussians, fibonacci_cameras  # noqa: E402

P, DEG, W, H = 400, 3, 80, 56


def scene():
    raw = make_gaussians(P, DEG, seed=77, scale_factor=0.9)
    cam = fibonacci_cameras(5, W, H, seed=78)[2]
    bg = torch.tensor([0.25, 0.5, 0.75])
    gc, gd = upstream_grads(H, W, seed=79)
    return raw, cam, bg, gc, gd


def main():
    raw, cam, bg, gc, gd = scene()
    out = {"bg": bg.numpy(), "gc": gc.numpy(), "gd": gd.numpy(),
           "viewmatrix": cam.world_view_transform.numpy(), "projmatrix": cam.full_proj_transform.numpy(),
           "campos": cam.camera_center.numpy(), "fov": np.array([cam.FoVx, cam.FoVy])}
    act = raw.activated()
    for k, v in act.items():
        out["in_" + k] = v.numpy()
    for aa in (0, 1):
        r = run_oracle(raw, cam, DEG, bg, torch.float64, antialiasing=bool(aa), gc=gc, gd=gd)
        out[f"aa{aa}_color"] = r["color"].numpy()
        out[f"aa{aa}_invdepth"] = r["invdepth"].numpy()
        out[f"aa{aa}_radii"] = r["radii"].numpy()
        out[f"aa{aa}_n_contrib"] = r["state"]["n_contrib"].numpy()
        out[f"aa{aa}_point_list"] = r["state"]["point_list"].numpy().astype(np.int32)
        out[f"aa{aa}_ranges"] = r["state"]["ranges"].numpy().astype(np.int32)
        for k, v in r["grads"].items():
            out[f"aa{aa}_grad_{k}"] = v.numpy()
    path = os.path.join(HERE, "oracle_small_scene.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
