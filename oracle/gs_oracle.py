"""CPU oracle for the differentiable Gaussian rasterizer path.   *** TEST INFRASTRUCTURE ONLY ***

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module.  The product path (`gaussian-splatting-slam_amd/`) never does: it fails loudly when the HIP
library is missing.

PARITY STATUS: **parity unpinned** for the rasterizer as a whole.  The arithmetic this file restates
lives in the third-party module `diff_gaussian_rasterization` (graphdeco-inria/diff-gaussian-rasterization
@ 9c5c2028f6fbee2be239bc4c9421ff894fe4fbe0, imported at reference `gaussian_renderer/__init__.py:14`);
its directory `submodules/diff-gaussian-rasterization/` is empty in /root/reference and the reference holds
no test, golden image or known-answer vector for it.  What IS pinned, by
fixtures generated from importable reference helpers (tests/golden/make_reference_fixtures.py):
  * SH -> RGB             against reference `utils/sh_utils.py:57-112` (eval_sh), deg 0..3
  * view / projection     against reference `utils/graphics_utils.py:38-71`, `scene/cameras.py:74-85`
  * psnr / l1 definitions against reference `utils/image_utils.py:17-19`, `utils/loss_utils.py:40-52`
Everything else follows the published algorithm of the pinned rasterizer as restated in SURVEY.md
Appendix A, and the boundary contract of reference `gaussian_renderer/__init__.py:18-121`.

Design: a vectorised pure-PyTorch forward whose gradients come from *autograd*, so the oracle's backward
is independent of the hand-derived backward the HIP kernels implement.  Three places where the published
rasterizer's backward deliberately differs from a literal autograd of its forward are encoded with
`.detach()` so that autograd reproduces the published behaviour:
  (q1) alpha = min(0.99, opacity*G) passes gradient straight through the clamp;
  (q2) the +-1.3*tanfov clamp of t.x/t.z, t.y/t.z zeroes d/dt.x (d/dt.y) and treats the clamped t.x (t.y)
       as a constant when differentiating J with respect to t.z;
  (q3) `scale_modifier` is applied inside the covariance (autograd handles it; the published code drops
       the factor in dL/dscale, which is only visible when scale_modifier != 1; we follow autograd).
Works in float32 or float64 (dtype of `means3D`); float64 gives the "true" answer used to calibrate the
fp32 tolerance written in the tests.
"""
from __future__ import annotations

import math
from typing import NamedTuple, Optional

import torch

BLOCK_X = 16
BLOCK_Y = 16

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
         -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435)


class OracleSettings(NamedTuple):
    """Same 13 fields, same order, as reference `gaussian_renderer/__init__.py:36-50`."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    antialiasing: bool


# --------------------------------------------------------------------------------------------------
# SH -> RGB   (basis, signs and ordering of reference utils/sh_utils.py:74-100)
# --------------------------------------------------------------------------------------------------
def sh_basis(deg: int, dirs: torch.Tensor) -> torch.Tensor:
    """[P,3] unit directions -> [P,(deg+1)^2] real SH basis values (constants folded in)."""
    x, y, z = dirs[:, 0], dirs[:, 1], dirs[:, 2]
    b = [torch.full_like(x, SH_C0)]
    if deg > 0:
        b += [-SH_C1 * y, SH_C1 * z, -SH_C1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [SH_C2[0] * xy, SH_C2[1] * yz, SH_C2[2] * (2.0 * zz - xx - yy), SH_C2[3] * xz,
              SH_C2[4] * (xx - yy)]
    if deg > 2:
        b += [SH_C3[0] * y * (3.0 * xx - yy), SH_C3[1] * xy * z, SH_C3[2] * y * (4.0 * zz - xx - yy),
              SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy), SH_C3[4] * x * (4.0 * zz - xx - yy),
              SH_C3[5] * z * (xx - yy), SH_C3[6] * x * (xx - 3.0 * yy)]
    return torch.stack(b, dim=1)


def sh_to_rgb(deg: int, shs: torch.Tensor, means3D: torch.Tensor, campos: torch.Tensor):
    """shs [P,M,3] (M >= (deg+1)^2).  Returns rgb[P,3] (clamped at 0) and clamped[P,3] bool.
    Follows reference gaussian_renderer/__init__.py:74-79 (colour = clamp_min(eval_sh + 0.5, 0))."""
    d = means3D - campos[None, :]
    d = d / d.norm(dim=1, keepdim=True)
    basis = sh_basis(deg, d)                                 # [P,K]
    k = basis.shape[1]
    rgb = (basis[:, :, None] * shs[:, :k, :]).sum(dim=1) + 0.5
    clamped = rgb < 0
    return torch.clamp_min(rgb, 0.0), clamped


# --------------------------------------------------------------------------------------------------
# covariance  (reference utils/general_utils.py:64-110, scene/gaussian_model.py:32-36)
# --------------------------------------------------------------------------------------------------
def quat_to_rotmat(q: torch.Tensor) -> torch.Tensor:
    """(w,x,y,z), used AS GIVEN (no normalisation inside the rasterizer)."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.view(-1, 3, 3)


def cov3d_from_scale_rot(scales, rotations, scale_modifier: float) -> torch.Tensor:
    """-> [P,6] (xx,xy,xz,yy,yz,zz) packing of reference utils/general_utils.py:64-73."""
    R = quat_to_rotmat(rotations)
    L = R * (scales * scale_modifier)[:, None, :]            # R @ diag(s)
    S = L @ L.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=1)


def _sym3(c6: torch.Tensor) -> torch.Tensor:
    return torch.stack([c6[:, 0], c6[:, 1], c6[:, 2],
                        c6[:, 1], c6[:, 3], c6[:, 4],
                        c6[:, 2], c6[:, 4], c6[:, 5]], dim=1).view(-1, 3, 3)


# --------------------------------------------------------------------------------------------------
# K1 preprocess
# --------------------------------------------------------------------------------------------------
class Preprocessed(NamedTuple):
    depths: torch.Tensor         # [P]   view-space z
    radii: torch.Tensor          # [P]   int32, 0 = culled
    xy: torch.Tensor             # [P,2] pixel coordinates of the mean
    conic: torch.Tensor          # [P,3] inverse 2-D covariance (A,B,C)
    opacity: torch.Tensor        # [P]   opacity * AA factor
    rgb: torch.Tensor            # [P,3]
    clamped: torch.Tensor        # [P,3] bool
    rect_min: torch.Tensor       # [P,2] int32 tile rect
    rect_max: torch.Tensor       # [P,2] int32
    tiles_touched: torch.Tensor  # [P]   int64
    cov3D: torch.Tensor          # [P,6]
    ndc: torch.Tensor            # [P,3] projected point (x,y,z)/(w+1e-7)


def preprocess(means3D, means2D, opacities, s: OracleSettings, shs=None, colors_precomp=None,
               scales=None, rotations=None, cov3D_precomp=None) -> Preprocessed:
    P = means3D.shape[0]
    dt = means3D.dtype
    W, H = int(s.image_width), int(s.image_height)
    vm = s.viewmatrix.to(dt)
    pm = s.projmatrix.to(dt)
    campos = s.campos.to(dt)

    # A.0/A.1: matrices arrive transposed (reference scene/cameras.py:69-71)
    t = means3D @ vm[:3, :3] + vm[3, :3][None, :]            # p_view
    hom = means3D @ pm[:3, :] + pm[3, :][None, :]            # [P,4]
    ndc = hom[:, :3] / (hom[:, 3:4] + 0.0000001)
    if means2D is not None:                                  # screen-space gradient side channel (§8b)
        ndc = ndc + means2D.to(dt) * torch.tensor([1.0, 1.0, 0.0], dtype=dt)
    in_front = t[:, 2] > 0.2

    # A.2 covariance
    if cov3D_precomp is not None:
        cov3D = cov3D_precomp
    else:
        cov3D = cov3d_from_scale_rot(scales, rotations, float(s.scale_modifier))
    Sigma = _sym3(cov3D)

    fx = W / (2.0 * s.tanfovx)
    fy = H / (2.0 * s.tanfovy)
    limx, limy = 1.3 * s.tanfovx, 1.3 * s.tanfovy
    tz = torch.where(in_front, t[:, 2], torch.ones_like(t[:, 2]))   # keep culled rows finite
    txtz, tytz = t[:, 0] / tz, t[:, 1] / tz
    in_x = (txtz >= -limx) & (txtz <= limx)
    in_y = (tytz >= -limy) & (tytz <= limy)
    # (q2): clamped value is a constant for the backward
    tx = torch.where(in_x, t[:, 0], (txtz.clamp(-limx, limx) * tz).detach())
    ty = torch.where(in_y, t[:, 1], (tytz.clamp(-limy, limy) * tz).detach())
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz),
                     zero, fy / tz, -(fy * ty) / (tz * tz)], dim=1).view(P, 2, 3)
    Wm = vm[:3, :3].t()                                      # rotation part of V
    M = J @ Wm[None]                                         # [P,2,3]
    cov2 = M @ Sigma @ M.transpose(1, 2)
    a0, b, c0 = cov2[:, 0, 0], cov2[:, 0, 1], cov2[:, 1, 1]

    # A.3 dilation / anti-aliasing
    det0 = a0 * c0 - b * b
    a, c = a0 + 0.3, c0 + 0.3
    det = a * c - b * b
    if s.antialiasing:
        h = torch.sqrt(torch.clamp_min(det0 / det, 0.000025))
    else:
        h = torch.ones_like(det)
    det_ok = det != 0
    det_s = torch.where(det_ok, det, torch.ones_like(det))
    conic = torch.stack([c / det_s, -b / det_s, a / det_s], dim=1)

    # A.4 extent, tile rect
    mid = 0.5 * (a + c)
    root = torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
    lam = torch.maximum(mid + root, mid - root)
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    xy = torch.stack([px, py], dim=1)
    gx, gy = (W + BLOCK_X - 1) // BLOCK_X, (H + BLOCK_Y - 1) // BLOCK_Y

    def _rect(p, r, block, grid, add):
        v = ((p + r * (1 if add else -1) + (block - 1 if add else 0)) / block).detach()
        v = torch.nan_to_num(v, nan=0.0, posinf=2.0e9, neginf=-2.0e9)
        return torch.trunc(v).clamp(0, grid).to(torch.int32)   # C-style (int) cast, then clamp
    rect_min = torch.stack([_rect(px, radius, BLOCK_X, gx, False), _rect(py, radius, BLOCK_Y, gy, False)], 1)
    rect_max = torch.stack([_rect(px, radius, BLOCK_X, gx, True), _rect(py, radius, BLOCK_Y, gy, True)], 1)
    area = ((rect_max[:, 0] - rect_min[:, 0]) * (rect_max[:, 1] - rect_min[:, 1])).to(torch.int64)
    visible = in_front & det_ok & (area > 0)
    radii = torch.where(visible, radius, torch.zeros_like(radius)).to(torch.int32)
    tiles = torch.where(visible, area, torch.zeros_like(area))

    if colors_precomp is not None:
        rgb = colors_precomp
        clamped = torch.zeros(P, 3, dtype=torch.bool)
    else:
        rgb, clamped = sh_to_rgb(int(s.sh_degree), shs, means3D, campos)
    op = opacities.reshape(P) * h
    return Preprocessed(t[:, 2], radii, xy, conic, op, rgb, clamped, rect_min, rect_max, tiles, cov3D, ndc)


# --------------------------------------------------------------------------------------------------
# K3-K5 binning: (tile, depth) sort, stable on emission order
# --------------------------------------------------------------------------------------------------
def _depth_bits(depths: torch.Tensor) -> torch.Tensor:
    """uint32 bit pattern of the fp32 depth as int64 (positive floats order like their bits)."""
    return depths.detach().to(torch.float32).view(torch.int32).to(torch.int64) & 0xFFFFFFFF


def bin_instances(pre: Preprocessed, W: int, H: int):
    """-> point_list[R] int64 (Gaussian ids sorted by (tile, depth bits), ties in emission order),
          ranges[tiles,2] int64, keys[R] int64 (sorted)."""
    gx, gy = (W + BLOCK_X - 1) // BLOCK_X, (H + BLOCK_Y - 1) // BLOCK_Y
    vis = torch.nonzero(pre.tiles_touched > 0).flatten()
    if vis.numel() == 0:
        e = torch.zeros(0, dtype=torch.int64)
        return e, torch.zeros(gx * gy, 2, dtype=torch.int64), e
    cnt = pre.tiles_touched[vis]
    gid = torch.repeat_interleave(vis, cnt)                  # emission order: ascending Gaussian id
    start = torch.cumsum(cnt, 0) - cnt
    local = torch.arange(gid.numel()) - torch.repeat_interleave(start, cnt)
    rw = (pre.rect_max[gid, 0] - pre.rect_min[gid, 0]).to(torch.int64)
    ty = pre.rect_min[gid, 1].to(torch.int64) + local // rw  # row-major inside the rect
    tx = pre.rect_min[gid, 0].to(torch.int64) + local % rw
    keys = ((ty * gx + tx) << 32) | _depth_bits(pre.depths)[gid]
    keys_sorted, perm = torch.sort(keys, stable=True)
    point_list = gid[perm]
    tile_of = keys_sorted >> 32
    ntiles = gx * gy
    counts = torch.bincount(tile_of, minlength=ntiles)
    ends = torch.cumsum(counts, 0)
    ranges = torch.stack([ends - counts, ends], dim=1)
    return point_list, ranges, keys_sorted


# --------------------------------------------------------------------------------------------------
# K6 alpha compositing (autograd provides K7)
# --------------------------------------------------------------------------------------------------
def composite(pre: Preprocessed, point_list, ranges, bg, W: int, H: int, tiles=None):
    """Front-to-back blending with the skip/stop rules of SURVEY A.5.
    `tiles`: optional iterable of tile ids to render (others keep bg / zeros) - used for full-size
    spot checks.  Returns color[3,H,W], invdepth[1,H,W], final_T[H,W], n_contrib[H,W] (int32)."""
    dt = pre.xy.dtype
    gx, gy = (W + BLOCK_X - 1) // BLOCK_X, (H + BLOCK_Y - 1) // BLOCK_Y
    bg = bg.to(dt)
    invd_all = 1.0 / torch.where(pre.radii > 0, pre.depths, torch.ones_like(pre.depths))
    color_tiles, depth_tiles, T_tiles, n_tiles, where = [], [], [], [], []
    tile_iter = range(gx * gy) if tiles is None else tiles
    for tile in tile_iter:
        ty, tx = divmod(int(tile), gx)
        x0, y0 = tx * BLOCK_X, ty * BLOCK_Y
        x1, y1 = min(x0 + BLOCK_X, W), min(y0 + BLOCK_Y, H)
        xs = torch.arange(x0, x1, dtype=dt)
        ys = torch.arange(y0, y1, dtype=dt)
        pxx = xs[None, :].expand(len(ys), len(xs)).reshape(-1)
        pyy = ys[:, None].expand(len(ys), len(xs)).reshape(-1)
        npx = pxx.numel()
        lo, hi = int(ranges[tile, 0]), int(ranges[tile, 1])
        where.append((y0, y1, x0, x1))
        if hi <= lo:
            color_tiles.append(bg[:, None].expand(3, npx))
            depth_tiles.append(torch.zeros(npx, dtype=dt))
            T_tiles.append(torch.ones(npx, dtype=dt))
            n_tiles.append(torch.zeros(npx, dtype=torch.int32))
            continue
        ids = point_list[lo:hi]
        xy, con, op = pre.xy[ids], pre.conic[ids], pre.opacity[ids]
        rgb, invd = pre.rgb[ids], invd_all[ids]
        dx = xy[:, 0:1] - pxx[None, :]
        dy = xy[:, 1:2] - pyy[None, :]
        power = -0.5 * (con[:, 0:1] * dx * dx + con[:, 2:3] * dy * dy) - con[:, 1:2] * dx * dy
        G = torch.exp(power)
        a_raw = op[:, None] * G
        alpha = a_raw + (torch.clamp_max(a_raw, 0.99) - a_raw).detach()          # (q1)
        valid = (power <= 0) & (alpha >= 1.0 / 255.0)
        a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
        T_incl = torch.cumprod(1.0 - a_eff, dim=0)
        T_before = torch.cat([torch.ones(1, npx, dtype=dt), T_incl[:-1]], dim=0)
        stop = valid & (T_incl < 0.0001)
        done = torch.cummax(stop.to(torch.int8), dim=0).values.bool()            # at/after first stop
        contrib = valid & ~done
        w = torch.where(contrib, a_eff * T_before, torch.zeros_like(a_eff))      # alpha * T
        C = (w[:, None, :] * rgb[:, :, None]).sum(dim=0)                         # [3,npx]
        D = (w * invd[:, None]).sum(dim=0)
        # final T = product over blended entries
        T_fin = torch.cumprod(torch.where(contrib, 1.0 - a_eff, torch.ones_like(a_eff)), dim=0)[-1]
        idx = torch.arange(1, hi - lo + 1, dtype=torch.int32)[:, None]
        last = torch.where(contrib, idx, torch.zeros_like(idx)).max(dim=0).values
        color_tiles.append(C + T_fin[None, :] * bg[:, None])
        depth_tiles.append(D)
        T_tiles.append(T_fin.detach())
        n_tiles.append(last.to(torch.int32))

    color = bg[:, None, None].expand(3, H, W).clone()
    invdepth = torch.zeros(1, H, W, dtype=dt)
    final_T = torch.ones(H, W, dtype=dt)
    n_contrib = torch.zeros(H, W, dtype=torch.int32)
    for (y0, y1, x0, x1), c, d, t, n in zip(where, color_tiles, depth_tiles, T_tiles, n_tiles):
        color[:, y0:y1, x0:x1] = c.view(3, y1 - y0, x1 - x0)
        invdepth[0, y0:y1, x0:x1] = d.view(y1 - y0, x1 - x0)
        final_T[y0:y1, x0:x1] = t.view(y1 - y0, x1 - x0)
        n_contrib[y0:y1, x0:x1] = n.view(y1 - y0, x1 - x0)
    return color, invdepth, final_T, n_contrib


# --------------------------------------------------------------------------------------------------
# Boundary: same call form as GaussianRasterizer.forward (reference gaussian_renderer/__init__.py:90-109)
# --------------------------------------------------------------------------------------------------
def rasterize(means3D, means2D, opacities, settings: OracleSettings, shs=None, colors_precomp=None,
              scales=None, rotations=None, cov3D_precomp=None, tiles=None, return_state=False):
    """Differentiable.  -> (color[3,H,W], radii[P] int32, invdepth[1,H,W]) (+ state dict)."""
    if (shs is None) == (colors_precomp is None):
        raise Exception('Please provide excatly one of either SHs or precomputed colors!')
    if ((scales is None or rotations is None) and cov3D_precomp is None) or \
            ((scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
    W, H = int(settings.image_width), int(settings.image_height)
    pre = preprocess(means3D, means2D, opacities, settings, shs, colors_precomp, scales, rotations,
                     cov3D_precomp)
    point_list, ranges, keys = bin_instances(pre, W, H)
    color, invdepth, final_T, n_contrib = composite(pre, point_list, ranges, settings.bg, W, H, tiles)
    if return_state:
        return color, pre.radii, invdepth, dict(pre=pre, point_list=point_list, ranges=ranges, keys=keys,
                                                final_T=final_T, n_contrib=n_contrib)
    return color, pre.radii, invdepth


def mark_visible(means3D, viewmatrix):
    """Near-plane test only."""
    vm = viewmatrix.to(means3D.dtype)
    z = means3D @ vm[:3, 2] + vm[3, 2]
    return z > 0.2


# --------------------------------------------------------------------------------------------------
# metric definitions (reference utils/image_utils.py:17-19, utils/loss_utils.py:40-52)
# --------------------------------------------------------------------------------------------------
def psnr(img1, img2):
    mse = ((img1 - img2) ** 2).view(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))


def l1_loss(a, b):
    return torch.abs(a - b).mean()
