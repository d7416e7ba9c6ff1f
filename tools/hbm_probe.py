"""What this box's HBM sustains for plain streaming kernels (torch elementwise ops), to put the Adam-carrying backward's
5 TB/s in context:  python tools/hbm_probe.py"""
import torch

def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

N = 1 << 28                     # 1 GiB per fp32 tensor
a = torch.empty(N, device="cuda").normal_()
b = torch.empty_like(a).normal_()
c = torch.empty_like(a)
print("copy     (1R+1W) %.2f TB/s" % (2 * 4 * N / t(lambda: c.copy_(a)) / 1e12))
print("add      (2R+1W) %.2f TB/s" % (3 * 4 * N / t(lambda: torch.add(a, b, out=c)) / 1e12))
print("inplace  (1R+1W) %.2f TB/s" % (2 * 4 * N / t(lambda: a.mul_(1.0001)) / 1e12))
print("fill     (0R+1W) %.2f TB/s" % (1 * 4 * N / t(lambda: c.fill_(1.0)) / 1e12))
print("sum      (1R+0W) %.2f TB/s" % (1 * 4 * N / t(lambda: a.sum()) / 1e12))
M = 59 * 1000000                # the shape of the C3 parameter set
p, m, v, g = (torch.empty(M, device="cuda").normal_() for _ in range(4))
def adam_like():
    torch._foreach_add_([p], [g])
print("p+=g at 236 MB (2R+1W) %.2f TB/s" % (3 * 4 * M / t(adam_like) / 1e12))
