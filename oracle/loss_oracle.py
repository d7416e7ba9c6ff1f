"""TEST INFRASTRUCTURE - CPU restatement of the reference's pure-PyTorch training loss, the oracle of csrc/ssim.hip.

Restates reference utils/loss_utils.py:40-52 (l1_loss) and :100-159 (ssim: 11x11 Gaussian window, sigma 1.5, zero padding,
C1 = 0.01^2, C2 = 0.03^2) and the combination of train.py:114-121 (the FUSED_SSIM_AVAILABLE == False branch).  Pinned against
the reference's own `ssim()` / `l1_loss()` outputs by tests/golden/reference_helpers.npz (`ssim_ab2`, `l1_ab`;
tests/test_oracle_golden.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product computes the loss in csrc/ssim.hip and has no CPU path.
"""
import math

import torch
import torch.nn.functional as F


def l1_loss(network_output, gt):
    return torch.abs(network_output - gt).mean()


_window_cache = {}


def _window(window_size, channel, like):
    key = (window_size, channel, like.device, like.dtype)
    w = _window_cache.get(key)
    if w is None:
        g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(window_size)])
        g = (g / g.sum()).unsqueeze(1)
        w2 = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)
        w = w2.expand(channel, 1, window_size, window_size).contiguous().to(device=like.device, dtype=like.dtype)
        _window_cache[key] = w
    return w


def ssim(img1, img2, window_size=11, size_average=True):
    channel = img1.size(-3)
    window = _window(window_size, channel, img1)
    pad = window_size // 2
    mu1 = F.conv2d(img1, window, padding=pad, groups=channel)
    mu2 = F.conv2d(img2, window, padding=pad, groups=channel)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(img1 * img1, window, padding=pad, groups=channel) - mu1_sq
    sigma2_sq = F.conv2d(img2 * img2, window, padding=pad, groups=channel) - mu2_sq
    sigma12 = F.conv2d(img1 * img2, window, padding=pad, groups=channel) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    if size_average:
        return ssim_map.mean()
    return ssim_map.mean(1).mean(1).mean(1)


def training_loss(image, gt_image, lambda_dssim=0.2):
    """reference train.py:114-121 with the pure-PyTorch ssim()."""
    Ll1 = l1_loss(image, gt_image)
    return (1.0 - lambda_dssim) * Ll1 + lambda_dssim * (1.0 - ssim(image, gt_image))
