"""`fused_ssim` module the reference tries first (train.py:31-35; used at :116-117 as
`fused_ssim(image.unsqueeze(0), gt_image.unsqueeze(0))`; submodule rahul-goel/fused-ssim, .gitmodules:10-12, absent from the
reference tree).  Forward + backward run in the HIP kernels of csrc/ssim.hip through the C ABI; value = mean of the SSIM
map the reference's pure-PyTorch `ssim()` (utils/loss_utils.py:100-159) computes.  Gradient flows to img1 only."""
import torch

from diff_gaussian_rasterization import _C

C1 = 0.01 ** 2
C2 = 0.03 ** 2


class FusedSSIMMap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, C1, C2, img1, img2, train=True):
        keep = bool(train and ctx.needs_input_grad[2])
        ssim_map, parts, i1, i2 = _C._ssim_forward(C1, C2, img1, img2, keep)
        if keep:
            ctx.save_for_backward(i1, i2, *parts)
        ctx.keep = keep
        return ssim_map

    @staticmethod
    def backward(ctx, opt_grad):
        if not ctx.keep:
            raise RuntimeError("fused_ssim was called with train=False; no backward state was kept")
        i1, i2, p0, p1, p2 = ctx.saved_tensors
        grad = _C.fusedssim_backward(None, None, i1, i2, opt_grad, partials=(p0, p1, p2))
        return None, None, grad, None, None


def fused_ssim(img1, img2, padding="same", train=True):
    if padding != "same":
        raise NotImplementedError("only padding='same' (zero padding, as the reference's ssim()) is implemented")
    return FusedSSIMMap.apply(C1, C2, img1, img2, train).mean()
