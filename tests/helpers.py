"""Shared helpers for the parity tests: run the HIP path through the reference-shaped Python boundary (which
calls the C ABI) and the CPU oracle on identical tensors."""
import ctypes as C
import math

import numpy as np
import torch

from oracle import gs_oracle as O
from scene_utils import make_gaussians, fibonacci_cameras, look_at_camera


def settings_for(cam, deg, bg, scale_modifier=1.0, antialiasing=False, cls=None, device="cpu", debug=False):
    kw = dict(image_height=int(cam.image_height), image_width=int(cam.image_width),
              tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg.to(device),
              scale_modifier=scale_modifier, viewmatrix=cam.world_view_transform.to(device),
              projmatrix=cam.full_proj_transform.to(device), sh_degree=deg, campos=cam.camera_center.to(device),
              prefiltered=False, debug=debug, antialiasing=antialiasing)
    return (cls or O.OracleSettings)(**kw)


def leaf_inputs(raw, dtype, device, mode="sh", scale_modifier=1.0):
    """Activated rasterizer inputs as fresh leaves.  mode: 'sh' | 'dc' | 'colors' ; returns dict"""
    act = raw.activated()
    d = {}
    for k in ("means3D", "opacities", "scales", "rotations"):
        d[k] = act[k].detach().to(device=device, dtype=dtype).clone().requires_grad_(True)
    shs = act["shs"].detach().to(device=device, dtype=dtype)
    if mode == "sh":
        d["shs"] = shs.clone().requires_grad_(True)
    elif mode == "dc":
        d["dc"] = shs[:, :1].clone().contiguous().requires_grad_(True)
        d["shs"] = shs[:, 1:].clone().contiguous().requires_grad_(True)
    elif mode == "colors":
        gen = torch.Generator().manual_seed(7)
        d["colors_precomp"] = torch.rand(shs.shape[0], 3, generator=gen).to(device=device, dtype=dtype).requires_grad_(True)
    d["means2D"] = torch.zeros(shs.shape[0], 3, device=device, dtype=dtype, requires_grad=True)
    return d


def upstream_grads(H, W, seed=3, depth=True):
    gen = torch.Generator().manual_seed(seed)
    gc = torch.randn(3, H, W, generator=gen)
    gd = torch.randn(1, H, W, generator=gen) if depth else torch.zeros(1, H, W)
    return gc, gd


def run_oracle(raw, cam, deg, bg, dtype=torch.float64, mode="sh", antialiasing=False, scale_modifier=1.0,
               gc=None, gd=None, cov_precomp=False, tiles=None):
    inp = leaf_inputs(raw, dtype, "cpu", mode)
    s = settings_for(cam, deg, bg, scale_modifier, antialiasing)
    kw = dict(shs=None, colors_precomp=inp.get("colors_precomp"))
    if mode == "sh":
        kw["shs"] = inp["shs"]
    elif mode == "dc":
        kw["shs"] = torch.cat([inp["dc"], inp["shs"]], dim=1)
    if cov_precomp:
        cov = O.cov3d_from_scale_rot(inp["scales"], inp["rotations"], scale_modifier).detach().clone().requires_grad_(True)
        inp["cov3D_precomp"] = cov
        kw.update(cov3D_precomp=cov)
    else:
        kw.update(scales=inp["scales"], rotations=inp["rotations"])
    color, radii, invd, st = O.rasterize(inp["means3D"], inp["means2D"], inp["opacities"], s, return_state=True,
                                         tiles=tiles, **kw)
    grads = None
    if gc is not None:
        loss = (color * gc.to(dtype)).sum() + (invd * gd.to(dtype)).sum()
        loss.backward()
        grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in inp.items()}
    return dict(color=color.detach(), radii=radii, invdepth=invd.detach(), state=st, grads=grads, inputs=inp)


def run_hip(raw, cam, deg, bg, mode="sh", antialiasing=False, scale_modifier=1.0, gc=None, gd=None,
            cov_precomp=False, debug=False, device="cuda", **call_kw):
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    inp = leaf_inputs(raw, torch.float32, device, mode)
    s = settings_for(cam, deg, bg, scale_modifier, antialiasing, cls=GaussianRasterizationSettings, device=device,
                     debug=debug)
    kw = dict(shs=inp.get("shs"), colors_precomp=inp.get("colors_precomp"), dc=inp.get("dc"))
    if cov_precomp:
        cov = O.cov3d_from_scale_rot(inp["scales"].detach().cpu().double(), inp["rotations"].detach().cpu().double(),
                                     scale_modifier).float().to(device).requires_grad_(True)
        inp["cov3D_precomp"] = cov
        kw.update(cov3D_precomp=cov)
    else:
        kw.update(scales=inp["scales"], rotations=inp["rotations"])
    rast = GaussianRasterizer(s)
    color, radii, invd = rast(means3D=inp["means3D"], means2D=inp["means2D"], opacities=inp["opacities"], **kw, **call_kw)
    grads = None
    if gc is not None:
        loss = (color * gc.to(device)).sum()
        if gd is not None:          # (gd=None: no gradient arrives on the inverse-depth image - the kernels' DEPTH = false forms)
            loss = loss + (invd * gd.to(device)).sum()
        loss.backward()
        grads = {k: (v.grad.detach().cpu() if v.grad is not None else torch.zeros_like(v).cpu()) for k, v in inp.items()}
    torch.cuda.synchronize()
    return dict(color=color.detach().cpu(), radii=radii.cpu(), invdepth=invd.detach().cpu(), grads=grads, inputs=inp)


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def frac_close(a, b, atol):
    return float(((a.double() - b.double()).abs() <= atol).double().mean())


# ---- low-level: call the C ABI directly and pull the opaque state apart (bit-exact integer checks) ----
def _view(state, ptr, count, dtype):
    """CPU copy of `count` elements of `dtype` at raw device pointer `ptr`, which lies inside torch uint8 tensor `state`."""
    nbytes = count * torch.tensor([], dtype=dtype).element_size()
    off = ptr - state.data_ptr()
    assert 0 <= off and off + nbytes <= state.numel(), (off, nbytes, state.numel())
    return state[off:off + nbytes].clone().view(dtype).cpu()


def lowlevel_forward(raw, cam, deg, bg, antialiasing=False, device="cuda"):
    """Returns dict with rec, depth bits, order, tiles_touched, rect, offsets, point_list, ranges, final_T,
    n_contrib, color, invdepth, radii, R - straight from the C ABI."""
    from diff_gaussian_rasterization import _C, GaussianRasterizationSettings
    from diff_gaussian_rasterization import _settings_struct, _gauss_struct, _stream
    lib = _C.lib()
    inp = leaf_inputs(raw, torch.float32, device, "sh")
    P = inp["means3D"].shape[0]
    H, W = cam.image_height, cam.image_width
    rs = settings_for(cam, deg, bg, 1.0, antialiasing, cls=GaussianRasterizationSettings, device=device)
    s, keep = _settings_struct(rs, device)
    t = {k: v.detach().contiguous() for k, v in inp.items()}
    g = _gauss_struct(P, t["means3D"], None, t["shs"], None, t["opacities"], t["scales"], t["rotations"], None)
    geom = torch.zeros(lib.gsr_geometry_state_bytes(P), dtype=torch.uint8, device=device)
    img = torch.zeros(lib.gsr_image_state_bytes(W, H), dtype=torch.uint8, device=device)
    radii = torch.zeros(P, dtype=torch.int32, device=device)
    color = torch.empty(3, H, W, device=device)
    invd = torch.empty(1, H, W, device=device)
    R = _C.check(lib.gsr_forward_prepare(C.byref(s), C.byref(g), _C.ptr(geom), geom.numel(), _C.ptr(radii), _stream()))
    binning = torch.zeros(lib.gsr_binning_state_bytes(P, W, H, R), dtype=torch.uint8, device=device)
    _C.check(lib.gsr_forward_render(C.byref(s), C.byref(g), _C.ptr(geom), _C.ptr(binning), binning.numel(), R,
                                    _C.ptr(img), img.numel(), _C.ptr(color), _C.ptr(invd), 1, _stream()))
    torch.cuda.synchronize()
    pv = [C.c_void_p() for _ in range(6)]
    lib.gsr_debug_geometry_views(_C.ptr(geom), P, *[C.byref(p) for p in pv], None)
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    out = dict(R=R, radii=radii.cpu(), color=color.cpu(), invdepth=invd.cpu())
    out["rec"] = _view(geom, pv[0].value, P * 12, torch.float32).view(P, 12)
    out["depth_keys_sorted"] = _view(geom, pv[1].value, P, torch.int32).numpy().view(np.uint32)
    out["order"] = _view(geom, pv[2].value, P, torch.int32).numpy().view(np.uint32)
    out["tiles_touched"] = _view(geom, pv[3].value, P, torch.int32).numpy().view(np.uint32)
    out["rect"] = _view(geom, pv[4].value, P * 4, torch.int16).numpy().view(np.uint16).reshape(P, 4)
    out["offsets"] = _view(geom, pv[5].value, P, torch.int32).numpy().view(np.uint32)
    pb = [C.c_void_p() for _ in range(2)]
    lib.gsr_debug_binning_views(_C.ptr(binning), W, H, R, C.byref(pb[0]), C.byref(pb[1]))
    out["point_list"] = _view(binning, pb[0].value, R, torch.int32).numpy().view(np.uint32)
    out["ranges"] = _view(binning, pb[1].value, tiles * 2, torch.int32).numpy().view(np.uint32).reshape(tiles, 2)
    pi = [C.c_void_p() for _ in range(2)]
    lib.gsr_debug_image_views(_C.ptr(img), W, H, C.byref(pi[0]), C.byref(pi[1]))
    out["final_T"] = _view(img, pi[0].value, W * H, torch.float32).view(H, W)
    out["n_contrib"] = _view(img, pi[1].value, W * H, torch.int32).view(H, W)
    pw = [C.c_void_p() for _ in range(3)]
    classes = lib.gsr_debug_walk_views(_C.ptr(img), W, H, C.byref(pw[0]), C.byref(pw[1]), C.byref(pw[2]))
    out["walk_cnt"] = _view(img, pw[0].value, classes, torch.int32).numpy().view(np.uint32)
    out["walk_list"] = _view(img, pw[1].value, classes * tiles, torch.int32).numpy().view(np.uint32).reshape(classes, tiles)
    out["walk_of_tile"] = _view(img, pw[2].value, tiles, torch.int32).numpy().view(np.uint32)
    return out
