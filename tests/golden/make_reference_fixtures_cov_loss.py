"""Generates tests/golden/reference_cov_loss.npz  (run ONCE in the build container; never on the GPU box).

Round-4 extension of make_reference_fixtures.py: the remaining sub-results of the hot path that the reference's own,
importable Python can pin.  Only DATA (seeded inputs + the reference's outputs) is written; no reference source travels.

  python tests/golden/make_reference_fixtures_cov_loss.py        # needs /root/reference

Pins
  (a) the 3-D covariance  Sigma = R diag(mod*s)^2 R^T  and its 6-vector packing (xx,xy,xz,yy,yz,zz):
      `build_covariance_from_scaling_rotation` of scene/gaussian_model.py:32-36 (restated here as the three calls it makes,
      because scene/gaussian_model.py itself cannot be imported - SyntaxError at :410) over the reference's
      `build_scaling_rotation` / `build_rotation` / `strip_symmetric` (utils/general_utils.py:64-110), plus the Jacobians
      d cov6 / d scale and d cov6 / d (raw quaternion) by autograd THROUGH THE REFERENCE FUNCTIONS.
  (b) d ssim(a, b) / da                                (utils/loss_utils.py:100-159, autograd, float64 and float32)
  (c) the training loss of train.py:114-121, `(1 - l) * l1_loss(a, b) + l * (1 - ssim(a, b))`, value and gradient.

The reference's functions hard-code `device="cuda"` (and `dtype=torch.float`) in their `torch.zeros(...)` calls; there is no
GPU in the build container, so while this script runs the module global `torch` of utils/general_utils is replaced by a
forwarding proxy whose `zeros` allocates on the CPU ("the generator may remap the functions' hard-coded
device").  Two variants are recorded: `cov6_f32` = the reference exactly as written (float32 buffers), and `cov6_f64` = the same
code with the proxy also promoting the buffers to float64, so that a float64 oracle can be compared to 1e-12.
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_cov_loss.npz")


class _TorchOnCpu:
    """Forwards every attribute to torch; `zeros` ignores a requested "cuda" device (and optionally promotes to float64)."""

    def __init__(self, promote):
        self._promote = promote

    def __getattr__(self, name):
        return getattr(torch, name)

    @property
    def float(self):
        return torch.float64 if self._promote else torch.float32

    def zeros(self, *a, **k):
        k["device"] = "cpu"
        if self._promote:
            k["dtype"] = torch.float64
        return torch.zeros(*a, **k)


def _reference_cov6(gu, scaling, scaling_modifier, rotation):
    # the body of build_covariance_from_scaling_rotation, scene/gaussian_model.py:32-36, over the reference's helpers
    L = gu.build_scaling_rotation(scaling_modifier * scaling, rotation)
    actual_covariance = L @ L.transpose(1, 2)
    return gu.strip_symmetric(actual_covariance)


def main():
    sys.path.insert(0, REF)
    import utils.general_utils as gu                                  # noqa: E402
    from utils.loss_utils import l1_loss, ssim                       # noqa: E402

    gen = torch.Generator().manual_seed(20241220 + 4)
    out = {}

    # --- (a) covariance ---------------------------------------------------------------------------------------------
    n = 300
    mods = np.array([1.0, 0.5, 2.3])
    s = torch.exp(-2.5 + 0.8 * torch.randn(n, 3, generator=gen, dtype=torch.float64))       # exp(log-scale), gaussian_model.py:38
    q = torch.randn(n, 4, generator=gen, dtype=torch.float64) * torch.exp(torch.randn(n, 1, generator=gen, dtype=torch.float64))
    q[0] = torch.tensor([1.0, 0.0, 0.0, 0.0], dtype=torch.float64)                           # identity
    q[1] = torch.tensor([0.0, 0.0, 0.0, -2.0], dtype=torch.float64)                          # pi about z, not unit length
    s[2] = torch.tensor([1e-4, 3.0, 0.02], dtype=torch.float64)                              # needle
    mod_of = torch.tensor(mods[np.arange(n) % 3], dtype=torch.float64)
    out["cov_scales"] = s.numpy()
    out["cov_quat_raw"] = q.numpy()
    out["cov_quat_unit"] = torch.nn.functional.normalize(q).numpy()                          # rotation_activation, gaussian_model.py:46
    out["cov_modifier"] = mod_of.numpy()

    real_torch = gu.torch
    try:
        gu.torch = _TorchOnCpu(promote=False)
        c32 = torch.empty(n, 6, dtype=torch.float32)
        for m in mods:
            sel = mod_of == m
            c32[sel] = _reference_cov6(gu, s[sel].float(), float(m), q[sel].float())
        out["cov6_f32"] = c32.numpy()

        gu.torch = _TorchOnCpu(promote=True)
        c64 = torch.empty(n, 6, dtype=torch.float64)
        Js = torch.empty(n, 6, 3, dtype=torch.float64)
        Jq = torch.empty(n, 6, 4, dtype=torch.float64)
        for m in mods:
            sel = torch.nonzero(mod_of == m).flatten()
            ss = s[sel].clone().requires_grad_(True)
            qq = q[sel].clone().requires_grad_(True)
            c = _reference_cov6(gu, ss, float(m), qq)
            c64[sel] = c.detach()
            for k in range(6):                       # rows are independent, so d(sum_i c[i,k]) gives every row's Jacobian row k
                gs, gq = torch.autograd.grad(c[:, k].sum(), (ss, qq), retain_graph=True)
                Js[sel, k] = gs
                Jq[sel, k] = gq
        out["cov6_f64"] = c64.numpy()
        out["cov6_dscale"] = Js.numpy()
        out["cov6_dquat_raw"] = Jq.numpy()
    finally:
        gu.torch = real_torch

    # --- (b), (c) SSIM gradient and the training loss --------------------------------------------------------------
    for tag, shape, noise in (("s", (3, 40, 52), 0.1), ("m", (3, 67, 45), 0.05), ("one", (1, 33, 35), 0.2)):
        a = torch.rand(*shape, generator=gen)
        b = (a + noise * torch.randn(*shape, generator=gen)).clamp(0, 1)
        out[f"loss_{tag}_a"] = a.numpy()
        out[f"loss_{tag}_b"] = b.numpy()
        for dt, dn in ((torch.float64, "f64"), (torch.float32, "f32")):
            x = a.to(dt).clone().requires_grad_(True)
            v = ssim(x, b.to(dt))                                     # utils/loss_utils.py:100-159
            (g,) = torch.autograd.grad(v, x)
            out[f"loss_{tag}_ssim_{dn}"] = np.array(v.item())
            if dt == torch.float64:                                   # (gradients: the float64 evaluation only - it is the truth)
                out[f"loss_{tag}_dssim_da_{dn}"] = g.numpy()
            for lam in (0.2, 0.5):                                    # arguments/__init__.py:86 lambda_dssim = 0.2
                x = a.to(dt).clone().requires_grad_(True)
                Ll1 = l1_loss(x, b.to(dt))                           # train.py:115
                loss = (1.0 - lam) * Ll1 + lam * (1.0 - ssim(x, b.to(dt)))   # train.py:121
                (g,) = torch.autograd.grad(loss, x)
                out[f"loss_{tag}_train_l{int(lam * 10)}_{dn}"] = np.array(loss.item())
                if dt == torch.float64:
                    out[f"loss_{tag}_dtrain_da_l{int(lam * 10)}_{dn}"] = g.numpy()

    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
