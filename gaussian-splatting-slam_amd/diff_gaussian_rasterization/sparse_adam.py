"""Optimizers backed by the one-launch HIP Adam kernel (csrc/adam.hip).

`SparseGaussianAdam(params, lr, eps).step(visibility, N)`: interface the reference uses at train.py:37-41,173-176 and
scene/gaussian_model.py:171-176 (rows of Gaussians with visibility False are untouched; no bias correction - the
behaviour of the absent `3dgs_accel` module).  `FusedAdam`: torch.optim.Adam semantics for the reference's default
optimizer (scene/gaussian_model.py:169-170), all parameter groups updated by ONE kernel launch."""
import ctypes as C

import torch

from . import _C


def _arrays(tensors_p, tensors_g, tensors_m, tensors_v, lrs):
    n = len(tensors_p)
    vp = (C.c_void_p * n)(*[t.data_ptr() for t in tensors_p])
    vg = (C.c_void_p * n)(*[t.data_ptr() for t in tensors_g])
    vm = (C.c_void_p * n)(*[t.data_ptr() for t in tensors_m])
    vv = (C.c_void_p * n)(*[t.data_ptr() for t in tensors_v])
    num = (C.c_int64 * n)(*[t.numel() for t in tensors_p])
    lr = (C.c_float * n)(*lrs)
    return n, vp, vg, vm, vv, num, lr


class _GsrAdamBase(torch.optim.Adam):
    def init_state(self):
        """Create the moments of every parameter now (on the current stream) instead of lazily at the first step."""
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)

    def _collect(self, only=None):
        """`only`: optional collection of group names; other groups are left untouched this call."""
        ps, gs, ms, vs, lrs, states = [], [], [], [], [], []
        for group in self.param_groups:
            if only is not None and group.get("name") not in only:
                continue
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise _C.GsrError("HIP Adam needs contiguous fp32 parameters on the HIP device (no CPU path)")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                ps.append(p.data); gs.append(p.grad.contiguous()); ms.append(st["exp_avg"]); vs.append(st["exp_avg_sq"])
                lrs.append(float(group["lr"])); states.append(st)
        return ps, gs, ms, vs, lrs, states


class FusedAdam(_GsrAdamBase):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params=params, lr=lr, betas=betas, eps=eps)

    @torch.no_grad()
    def step(self, only=None):
        ps, gs, ms, vs, lrs, states = self._collect(only)
        if not ps:
            return
        g0 = self.param_groups[0]
        b1, b2 = g0["betas"]
        lib = _C.lib()
        for i in range(0, len(ps), 8):
            sl = slice(i, i + 8)
            for st in states[sl]:
                st["step"] += 1
            n, vp, vg, vm, vv, num, lr = _arrays(ps[sl], gs[sl], ms[sl], vs[sl], lrs[sl])
            steps = (C.c_int64 * n)(*[int(st["step"]) for st in states[sl]])
            with _C.on_device(ps[0].device):
                _C.check(lib.gsr_adam_step(n, vp, vg, vm, vv, num, lr, steps, float(b1), float(b2), float(g0["eps"]),
                                           _C._stream()))


class SparseGaussianAdam(_GsrAdamBase):
    def __init__(self, params, lr, eps):
        super().__init__(params=params, lr=lr, eps=eps)

    @torch.no_grad()
    def step(self, visibility, N, only=None):
        ps, gs, ms, vs, lrs, states = self._collect(only)
        if not ps:
            return
        vis = visibility.to(torch.uint8).contiguous()
        g0 = self.param_groups[0]
        lib = _C.lib()
        for i in range(0, len(ps), 8):
            sl = slice(i, i + 8)
            n, vp, vg, vm, vv, num, lr = _arrays(ps[sl], gs[sl], ms[sl], vs[sl], lrs[sl])
            with _C.on_device(ps[0].device):
                _C.check(lib.gsr_sparse_adam_step(n, vp, vg, vm, vv, num, lr, int(N), _C.ptr(vis), 0.9, 0.999,
                                                  float(g0["eps"]), _C._stream()))
