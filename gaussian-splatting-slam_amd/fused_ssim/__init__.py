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


class _FusedL1SSIMLoss(torch.autograd.Function):
    """(1-l) * mean|img1-img2| + l * (1 - mean(ssim_map(img1, img2))): reference train.py:114-121 in three HIP launches
    (forward + finalize, backward), no host synchronisation."""

    @staticmethod
    def forward(ctx, img1, img2, lambda_dssim):
        if not img1.is_cuda:
            raise _C.GsrError("fused_l1_ssim_loss needs tensors on the HIP device (no CPU path)")
        lib = _C.lib()
        a = img1.detach().float().contiguous()
        b = img2.detach().float().contiguous()
        H, W = int(a.shape[-2]), int(a.shape[-1])
        planes = a.numel() // (H * W)
        parts = [torch.empty_like(a) for _ in range(3)]
        nblk = int(lib.gsr_fused_loss_blocks(planes, H, W))
        partials = torch.empty(nblk, 2, dtype=torch.float32, device=a.device)
        loss = torch.empty((), dtype=torch.float32, device=a.device)
        with _C.on_device(a.device):
            _C.check(lib.gsr_fused_l1_ssim_forward(planes, H, W, C1, C2, float(lambda_dssim), _C.ptr(a), _C.ptr(b),
                                                   _C.ptr(parts[0]), _C.ptr(parts[1]), _C.ptr(parts[2]),
                                                   _C.ptr(partials), _C.ptr(loss), _C._stream()))
        ctx.save_for_backward(a, b, *parts)
        ctx.lam = float(lambda_dssim)
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = _C.lib()
        a, b, p0, p1, p2 = ctx.saved_tensors
        H, W = int(a.shape[-2]), int(a.shape[-1])
        planes = a.numel() // (H * W)
        g = g.detach().float().contiguous()                # device scalar dL/dloss: read by the kernel, never by the host
        out = torch.empty_like(a)
        with _C.on_device(a.device):
            _C.check(lib.gsr_fused_l1_ssim_backward(planes, H, W, ctx.lam, _C.ptr(a), _C.ptr(b), _C.ptr(g), _C.ptr(p0),
                                                    _C.ptr(p1), _C.ptr(p2), _C.ptr(out), _C._stream()))
        return out, None, None


def fused_l1_ssim_loss(img1, img2, lambda_dssim=0.2):
    return _FusedL1SSIMLoss.apply(img1, img2, lambda_dssim)


class _L1Mean(torch.autograd.Function):
    """weight * mean|(a - b) mask|, gradient to `a`: the inverse-depth term of reference train.py:124-132 in three HIP launches."""

    @staticmethod
    def forward(ctx, a, b, weight, mask):
        if not a.is_cuda:
            raise _C.GsrError("l1_mean_loss needs tensors on the HIP device (no CPU path)")
        lib = _C.lib()
        x = a.detach().float().contiguous()
        y = b.detach().float().contiguous()
        m = mask.detach().float().expand_as(x).contiguous() if mask is not None else None
        if x.shape != y.shape:
            raise ValueError("l1_mean_loss: shapes differ")
        partials = torch.empty(int(lib.gsr_l1_mean_blocks()), dtype=torch.float32, device=x.device)
        out = torch.empty((), dtype=torch.float32, device=x.device)
        with _C.on_device(x.device):
            _C.check(lib.gsr_l1_mean_forward(x.numel(), float(weight), _C.ptr(x), _C.ptr(y), _C.ptr(m) if m is not None else None,
                                             _C.ptr(partials), _C.ptr(out), _C._stream()))
        ctx.save_for_backward(x, y, *([m] if m is not None else []))
        ctx.weight = float(weight)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _C.lib()
        x, y, *m = ctx.saved_tensors
        g = g.detach().float().contiguous()
        grad = torch.empty_like(x)
        with _C.on_device(x.device):
            _C.check(lib.gsr_l1_mean_backward(x.numel(), ctx.weight, _C.ptr(x), _C.ptr(y), _C.ptr(m[0]) if m else None, _C.ptr(g),
                                              _C.ptr(grad), _C._stream()))
        return grad, None, None, None


def l1_mean_loss(a, b, weight=1.0, mask=None):
    """`weight * torch.abs((a - b) * mask).mean()` (reference train.py:130-131), gradient to `a`."""
    return _L1Mean.apply(a, b, weight, mask)
