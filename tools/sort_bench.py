#!/usr/bin/env python3
"""Times the library's radix sort in isolation (gsr_debug_radix_sort): the two sorts of a C3 forward.
   GSR_LIB=<variant .so> python tools/sort_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-slam_amd"))
import numpy as np, torch
from diff_gaussian_rasterization import _C
lib = _C.lib()
dev = "cuda"
def bench(n, cap, bits, dual, label, reps=40):
    rng = np.random.default_rng(1)
    if bits == 32:   # depth-like keys: floats in [1.75, 6.25]
        kn = rng.uniform(1.75, 6.25, cap).astype(np.float32).view(np.uint32)
    else:
        kn = rng.integers(0, 8160, cap, dtype=np.uint64).astype(np.uint32)
    src = torch.from_numpy(kn.view(np.int32).copy()).to(dev)
    k0 = torch.empty_like(src); k1 = torch.empty_like(src)
    v0 = torch.empty(cap, dtype=torch.int32, device=dev); v1 = torch.empty_like(v0)
    w0 = torch.arange(cap, dtype=torch.int32, device=dev) if dual else None
    w1 = torch.empty_like(v0) if dual else None
    tmp = torch.empty(lib.gsr_debug_radix_tmp_bytes(cap), dtype=torch.uint8, device=dev)
    nd = torch.tensor([n, 0], dtype=torch.int32, device=dev) if n != cap else None
    ts = []
    for i in range(reps + 5):
        k0.copy_(src)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _C.check(lib.gsr_debug_radix_sort(_C.ptr(k0), _C.ptr(v0), _C.ptr(k1), _C.ptr(v1), _C.ptr(w0), _C.ptr(w1), cap, bits, 1,
                                          _C.ptr(nd), _C.ptr(tmp), _C._stream()))
        b.record(); torch.cuda.synchronize()
        if i >= 5: ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    print(f"{label:34s} n={n:8d} cap={cap:8d} bits={bits:2d} dual={int(dual)}  median {ts[len(ts)//2]:7.1f} us  min {ts[0]:7.1f} us")
bench(1_000_000, 1_000_000, 32, False, "depth sort (C3)")
bench(4_500_000, 5_625_000, 13, True, "tile sort (C3, capacity 1.25x)")
bench(4_500_000, 4_500_000, 13, True, "tile sort (C3, exact)")
bench(4_500_000, 5_625_000, 13, False, "tile sort fwd-only")
bench(100_000, 100_000, 32, False, "depth sort (C2)")
bench(700_000, 875_000, 12, True, "tile sort (C5)")
