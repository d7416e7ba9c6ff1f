"""Error of the two reductions of the one-wave-per-tile compositing backward (matrix pipe = default, GSR_BWD_REDUCE=swap = the
v_permlane / DPP tree) against the float64 oracle, per gradient tensor: rel-L2 and max-abs / max|g|.  GPU box, repo root:
    python tests/sweeps/mx_vs_swap_error.py      (uses the CPU oracle like the tests do; not collected by pytest)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-slam_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from helpers import make_gaussians, fibonacci_cameras, upstream_grads, run_hip, run_oracle   # noqa: E402

os.environ["GSR_BWD_FORM"] = "tile"
for name, P, sh, aniso, scale, depth, aa in (("regular", 6000, 2, 0.0, 0.8, False, False), ("depth+aa", 6000, 2, 0.0, 0.8, True, True),
                                           ("needles", 6000, 2, 1.3, 0.8, False, False), ("big splats", 3000, 1, 0.0, 3.0, False, False),
                                           ("big needles", 3000, 1, 1.0, 3.0, False, False)):
    raw = make_gaussians(P, sh, seed=401, scale_factor=scale)
    if aniso:
        gen = torch.Generator().manual_seed(402)
        raw.scaling = raw.scaling + aniso * torch.randn(raw.scaling.shape, generator=gen)
    cam = fibonacci_cameras(3, 208, 144, seed=403)[2]
    bg = torch.tensor([0.3, 0.2, 0.1])
    gc, gd = upstream_grads(cam.image_height, cam.image_width, depth=depth)
    ref = run_oracle(raw, cam, sh, bg, torch.float64, antialiasing=aa, gc=gc, gd=gd)
    res = {}
    for red in ("swap", "mfma"):
        os.environ["GSR_BWD_REDUCE"] = red
        res[red] = run_hip(raw, cam, sh, bg, antialiasing=aa, gc=gc, gd=gd if depth else None)
    print(f"== {name}")
    for k in res["swap"]["grads"]:
        r = ref["grads"][k].double()
        row = []
        for red in ("swap", "mfma"):
            g = res[red]["grads"][k].double().cpu()
            row.append("%s rel-L2 %.2e max %.2e" % (red, float((g - r).norm() / (r.norm() + 1e-300)), float((g - r).abs().max() / (r.abs().max() + 1e-300))))
        d = (res["swap"]["grads"][k].double() - res["mfma"]["grads"][k].double()).cpu()
        print("  %-12s %s | %s | swap-mfma rel-L2 %.2e" % (k, row[0], row[1], float(d.norm() / (r.norm() + 1e-300))))
