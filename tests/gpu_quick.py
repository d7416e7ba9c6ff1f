"""Diagnostic (not a pytest): HIP vs oracle on a small scene, prints error statistics."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa
import torch, numpy as np
from helpers import *
from scene_utils import make_gaussians, fibonacci_cameras

def main():
    P, deg, W, H = int(os.environ.get("P", 3000)), 3, 150, 100
    raw = make_gaussians(P, deg, seed=11, scale_factor=0.6)
    cam = fibonacci_cameras(3, W, H, seed=5)[1]
    bg = torch.tensor([0.2, 0.5, 0.7])
    for aa in (False, True):
        gc, gd = upstream_grads(H, W)
        t0 = time.time()
        ref = run_oracle(raw, cam, deg, bg, torch.float64, antialiasing=aa, gc=gc, gd=gd)
        t1 = time.time()
        out = run_hip(raw, cam, deg, bg, antialiasing=aa, gc=gc, gd=gd, debug=True)
        print(f"--- antialiasing={aa}  oracle {t1-t0:.1f}s  R={ref['state']['point_list'].numel()} visible={(ref['radii']>0).sum().item()}")
        print(" radii mismatch:", int((ref["radii"] != out["radii"]).sum()), "of", P)
        dc = (ref["color"] - out["color"].double()).abs()
        print(" color max-abs %.3e  frac<=2e-5 %.6f  mean %.3e" % (dc.max(), (dc <= 2e-5).double().mean(), dc.mean()))
        dd = (ref["invdepth"] - out["invdepth"].double()).abs()
        print(" invdepth max-abs %.3e frac<=2e-5 %.6f" % (dd.max(), (dd <= 2e-5).double().mean()))
        for k in ref["grads"]:
            a, b = out["grads"][k], ref["grads"][k]
            print("  grad %-10s rel_l2 %.3e  max-abs %.3e (max|g| %.3e)" % (k, rel_l2(a, b), (a.double()-b.double()).abs().max(), b.abs().max()))
    ll = lowlevel_forward(raw, cam, deg, bg)
    print("lowlevel R", ll["R"], "n_contrib max", ll["n_contrib"].max().item())
    st = ref["state"]
    print(" n_contrib mismatch", int((ll["n_contrib"] != st["n_contrib"]).sum()), " final_T maxdiff", float((ll["final_T"].double()-st["final_T"]).abs().max()))
    # bit-exact binning check against keys built from the GPU's own depth bits / rects
    order = ll["order"]; tt = ll["tiles_touched"]; rect = ll["rect"].astype(np.int64)
    depth_bits = ll["rec"][:, 11].numpy().view(np.uint32).astype(np.uint64)
    gx = (W + 15) // 16
    keys = []; ids = []
    for g in range(P):
        if tt[g] == 0: continue
        x0, y0, x1, y1 = rect[g]
        for y in range(y0, y1):
            for x in range(x0, x1):
                keys.append((np.uint64(y * gx + x) << np.uint64(32)) | depth_bits[g]); ids.append(g)
    keys = np.array(keys, dtype=np.uint64); ids = np.array(ids)
    perm = np.argsort(keys, kind="stable")
    exp_list = ids[perm]
    print(" point_list exact:", bool(len(exp_list) == ll["R"] and (exp_list == ll["point_list"]).all()))
    tile_sorted = (keys[perm] >> np.uint64(32)).astype(np.int64)
    cnt = np.bincount(tile_sorted, minlength=ll["ranges"].shape[0]); ends = np.cumsum(cnt)
    exp_ranges = np.stack([ends - cnt, ends], 1); exp_ranges[cnt == 0] = 0
    print(" ranges exact:", bool((exp_ranges == ll["ranges"]).all()))
    print(" oracle point_list equal:", bool((st["point_list"].numpy() == ll["point_list"]).all()) if len(st["point_list"]) == ll["R"] else "len differs")

if __name__ == "__main__":
    main()
