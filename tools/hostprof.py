import cProfile, pstats, sys, os, io
ROOT="/root/repo"
sys.path.insert(0, os.path.join(ROOT,"gaussian-splatting-slam_amd")); sys.path.insert(0, ROOT)
import torch
from scene_utils import make_config, GaussianModel, Trainer
from gaussian_renderer import render, PipelineParams
raw, cams, cfg = make_config(1, views=4)
for c in cams: c.to("cuda")
pipe=PipelineParams(); bg=torch.zeros(3,device="cuda")
model=GaussianModel.from_raw(raw.to("cuda"))
gts={i: torch.rand(3,cfg["H"],cfg["W"],device="cuda") for i in range(4)}
tr=Trainer(model,cams,gts,render,pipe,bg,separate_sh=True,optimizer=sys.argv[1] if len(sys.argv)>1 else "hip_fused")
for i in range(20): tr.step(i%4)
torch.cuda.synchronize()
pr=cProfile.Profile(); pr.enable()
for i in range(300): tr.step(i%4)
pr.disable(); torch.cuda.synchronize()
s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
