"""Fused parameter activations of the Gaussian model (reference scene/gaussian_model.py:38-46, :101-121): exp on the
scales, normalize on the quaternions, sigmoid on the opacities - one HIP launch forward and one backward
(`gsr_gaussian_activations_*` in include/gsr.h) instead of the ~25 small PyTorch launches the three separate ops and their
autograd graphs cost per training step.  Device tensors only; there is no CPU fallback here - the plain torch getters of
`GaussianModel` serve CPU tensors."""
import torch

from diff_gaussian_rasterization import _C


class _GaussianActivations(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw_scaling, raw_rotation, raw_opacity):
        P = raw_scaling.shape[0]
        if not (raw_scaling.is_cuda and raw_rotation.is_cuda and raw_opacity.is_cuda):
            raise RuntimeError("gaussian_activations: parameters must live on the HIP device")
        if raw_scaling.shape != (P, 3) or raw_rotation.shape != (P, 4) or raw_opacity.numel() != P:
            raise ValueError("gaussian_activations: expected [P,3] scales, [P,4] rotations, [P,1] opacities")
        rs, rq, ro = (t.detach().contiguous().float() for t in (raw_scaling, raw_rotation, raw_opacity))
        s, q, o = torch.empty_like(rs), torch.empty_like(rq), torch.empty_like(ro)
        _C.check(_C.lib().gsr_gaussian_activations_forward(P, _C.ptr(rs), _C.ptr(rq), _C.ptr(ro), _C.ptr(s), _C.ptr(q),
                                                           _C.ptr(o), _C._stream()))
        ctx.save_for_backward(rq, s, o)
        ctx.set_materialize_grads(False)
        return s, q, o

    @staticmethod
    def backward(ctx, g_s, g_q, g_o):
        rq, s, o = ctx.saved_tensors
        P = s.shape[0]
        g_s, g_q, g_o = (None if g is None else g.contiguous().float() for g in (g_s, g_q, g_o))
        d_s, d_q, d_o = torch.empty_like(s), torch.empty_like(rq), torch.empty_like(o)
        _C.check(_C.lib().gsr_gaussian_activations_backward(
            P, _C.ptr(rq), _C.ptr(s), _C.ptr(o), _C.ptr(g_s), _C.ptr(g_q), _C.ptr(g_o), _C.ptr(d_s), _C.ptr(d_q),
            _C.ptr(d_o), _C._stream()))
        return d_s, d_q, d_o


def gaussian_activations(raw_scaling, raw_rotation, raw_opacity):
    """(exp(raw_scaling), normalize(raw_rotation), sigmoid(raw_opacity)), differentiable."""
    return _GaussianActivations.apply(raw_scaling, raw_rotation, raw_opacity)
