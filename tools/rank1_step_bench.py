#!/usr/bin/env python3
"""What the view-sharded step costs a rank on ITS OWN GPU at N = 8 (everything but the collectives), BASELINE configs[2]:
   (a) exchange "allreduce": backward (all gradients) + one-launch Adam over the six groups;
   (b) exchange "sh_rank1", rebuild -> .grad -> Adam: backward (all gradients) + gsr_sh_rank1_expand + Adam;
   (c) exchange "sh_rank1" as shipped: backward WITHOUT dL/df_rest + gsr_sh_rank1_adam (SH groups) + Adam (geometry groups).
The gathered buffer is synthetic (this rank's dL/df_dc replicated 8 times with different camera centres): same bytes, same work.
    python tools/rank1_step_bench.py            (GPU box, repo root)"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-slam_amd"))
import torch  # noqa: E402


def main():
    from scene_utils import make_config, GaussianModel
    from scene_utils.parallel import _fused_sh_adam_args
    from scene_utils.losses import training_loss_fused
    from gaussian_renderer import render, PipelineParams
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _C
    lib = _C.lib()
    dev = "cuda"
    N = 8
    raw, cams, cfg = make_config(3, views=8)
    for c in cams:
        c.to(dev)
    pipe, bg = PipelineParams(), torch.zeros(3, device=dev)
    with torch.no_grad():
        teacher = GaussianModel.from_raw(raw.to(dev), requires_grad=False)
        gt = render(cams[0], teacher, pipe, bg)["render"].clamp(0, 1).clone()
    out = {}
    for variant in ("allreduce", "rank1_unfused", "rank1_fused"):
        model = GaussianModel.from_raw(raw.to(dev), requires_grad=True)
        opt = model.training_setup(optimizer="hip")
        P = model.get_xyz.shape[0]
        lib.gsr_profile_enable(1)
        for it in range(8):
            if it == 3:
                torch.cuda.synchronize()
                lib.gsr_profile_reset()
                t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
                t0.record()
            fold = dgr.BackwardFold(skip_sh_rest=True) if variant == "rank1_fused" else None
            pkg = render(cams[it % 8], model, pipe, bg, separate_sh=True, fold=fold)
            loss = training_loss_fused(pkg["render"], gt, 0.2)
            loss.backward()
            with torch.no_grad():
                if variant != "allreduce":
                    g = model._features_dc.grad.reshape(P, 3)
                    gathered = torch.empty(N, P + 1, 3, device=dev)
                    gathered[:, :P] = g
                    for r in range(N):
                        gathered[r, P] = cams[r].camera_center
                    if variant == "rank1_fused":
                        fa, keep = _fused_sh_adam_args(opt, model._features_dc, model._features_rest)
                        _C.check(lib.gsr_sh_rank1_adam(P, N, 3, 15, _C.ptr(model._xyz.detach()), _C.ptr(gathered), C.c_float(1.0 / N),
                                                       _C.ptr(model._features_dc.data), _C.ptr(model._features_rest.data),
                                                       C.byref(fa), _C._stream()))
                        model._features_dc.grad = None
                        model._features_rest.grad = None
                    else:
                        od, orr = torch.empty_like(model._features_dc), torch.empty_like(model._features_rest)
                        _C.check(lib.gsr_sh_rank1_expand(P, N, 3, 15, _C.ptr(model._xyz.detach()), _C.ptr(gathered), C.c_float(1.0 / N),
                                                         _C.ptr(od), _C.ptr(orr), _C._stream()))
                        model._features_dc.grad, model._features_rest.grad = od, orr
                opt.step()
                opt.zero_grad(set_to_none=True)
        t1.record()
        torch.cuda.synchronize()
        prof = _C.profile_read()
        lib.gsr_profile_enable(0)
        keep_k = ("preprocess_bwd", "adam_dense", "sh_rank1_expand", "sh_rank1_adam", "render_bwd")
        out[variant] = {"ms_per_step_gpu": round(t0.elapsed_time(t1) / 5, 4),
                        "kernels_ms": {k: round(v[0] / v[1], 4) for k, v in prof.items() if k in keep_k}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
