"""Diagnostic: forward-only render of every view of a config, with per-view progress (not a pytest)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa
import torch
from scene_utils import make_config, GaussianModel
from gaussian_renderer import render, PipelineParams
import diff_gaussian_rasterization as dgr
cfg_id, views = int(sys.argv[1]), int(sys.argv[2])
raw, cams, c = make_config(cfg_id, views=views)
print("generated", c, flush=True)
model = GaussianModel.from_raw(raw.to("cuda"), requires_grad=False)
bg = torch.zeros(3, device="cuda")
for i, cam in enumerate(cams):
    cam.to("cuda")
    t0 = time.time()
    with torch.no_grad():
        pkg = render(cam, model, PipelineParams(), bg)
    torch.cuda.synchronize()
    print(f"view {i}: R={dgr.call_stats()["num_rendered"]} visible={int(pkg['visibility_filter'].sum())} "
          f"max radius={int(pkg['radii'].max())} {1e3*(time.time()-t0):.1f} ms", flush=True)
