#!/usr/bin/env python3
"""Extended sweep of the fused loss kernels (GPU box, repo root):  python tests/sweeps/extended_loss_sweep.py [first] [count]
Random image sizes from 1x1 up (narrower than the 11-tap window included), 1-4 planes, random lambda: fused_l1_ssim_loss,
fused_ssim (map form) and l1_mean_loss - value and gradient - against the pure-PyTorch restatement in float64
(oracle/loss_oracle.py = reference utils/loss_utils.py:100-159 + train.py:114-121)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-slam_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

torch.set_num_threads(16)      # (the box's CPU share; torch's default there is 128 threads on a 16-CPU quota)

from fused_ssim import fused_l1_ssim_loss, fused_ssim, l1_mean_loss  # noqa: E402
from oracle import loss_oracle as LO  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad, t0 = [], time.time()
for seed in range(first, first + count):
    g = torch.Generator().manual_seed(2000 + seed)

    def u(a, b):
        return a + (b - a) * float(torch.rand((), generator=g))
    C = int(u(1, 5))
    H, W = (int(u(1, 14)), int(u(1, 14))) if seed % 5 == 0 else (int(u(1, 260)), int(u(1, 330)))
    lam = u(0.0, 1.0)
    a = torch.rand(C, H, W, generator=g, dtype=torch.float64)
    b = (a + u(0.01, 0.3) * torch.randn(C, H, W, generator=g, dtype=torch.float64)).clamp(0, 1)
    try:
        a_ref = a.clone().requires_grad_(True)
        ref = LO.training_loss(a_ref, b, lam)
        (1.7 * ref).backward()
        x = a.float().cuda().requires_grad_(True)
        val = fused_l1_ssim_loss(x, b.float().cuda(), lam)
        (1.7 * val).backward()
        assert abs(val.item() - ref.item()) <= 3e-6, ("loss value", val.item(), ref.item())
        gr, gg = a_ref.grad, x.grad.cpu().double()
        far = (a - b).abs() > 1e-6               # (the L1 term's sign is decided in fp32)
        err = (gr - gg).abs()[far]
        assert float(err.max() if err.numel() else 0.0) <= 2e-5 * max(1e-3, float(gr.abs().max())) + 1e-9, ("loss grad", float(err.max()))
        # map form
        a2 = a.clone().requires_grad_(True)
        m_ref = LO.ssim(a2.unsqueeze(0), b.unsqueeze(0))
        m_ref.backward()
        y = a.float().cuda().requires_grad_(True)
        m = fused_ssim(y.unsqueeze(0), b.float().cuda().unsqueeze(0))
        m.backward()
        assert abs(m.item() - m_ref.item()) <= 3e-6, ("ssim value", m.item(), m_ref.item())
        e2 = (a2.grad - y.grad.cpu().double()).abs()
        assert float(e2.max()) <= 2e-5 * max(1e-3, float(a2.grad.abs().max())) + 1e-9, ("ssim grad", float(e2.max()))
        # masked L1 mean
        mask = (torch.rand(C, H, W, generator=g) > 0.4).double()
        a3 = a.clone().requires_grad_(True)
        l_ref = 0.6 * torch.abs((a3 - b) * mask).mean()
        l_ref.backward()
        z = a.float().cuda().requires_grad_(True)
        lv = l1_mean_loss(z, b.float().cuda(), 0.6, mask.float().cuda())
        lv.backward()
        assert abs(lv.item() - l_ref.item()) <= 2e-6 * max(1.0, abs(l_ref.item())), ("l1 value", lv.item(), l_ref.item())
        assert torch.allclose(z.grad.cpu().double()[far], a3.grad[far], rtol=1e-5, atol=1e-12), "l1 grad"
    except Exception as e:      # noqa: BLE001
        bad.append(seed)
        print(f"seed {seed} ({C}x{H}x{W}, lambda {lam:.2f}): {type(e).__name__}: {str(e)[:200]}", flush=True)
    if (seed - first) % 50 == 49:
        print(f"... {seed - first + 1} cases, {len(bad)} failures, {time.time() - t0:.0f} s", flush=True)
print(f"loss sweep: seeds {first}..{first + count - 1}: {count - len(bad)} passed, {len(bad)} failed {bad}")
sys.exit(1 if bad else 0)
