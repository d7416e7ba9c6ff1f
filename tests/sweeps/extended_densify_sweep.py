#!/usr/bin/env python3
"""Extended sweep of densify_and_prune against its CPU restatement (GPU box, repo root):
    python tests/sweeps/extended_densify_sweep.py [first] [count]
Random model sizes (1 .. 30 000 rows, odd counts), scale spreads, gradient thresholds, extents, screen-size limits, with and
without optimizer moments: the rows kept / cloned / split, their order, every parameter and moment of kept and cloned rows and
everything but the sampled position of split rows must equal oracle/densify_oracle.py exactly (tests/test_densify_gpu.py checks
one fixed scene)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-slam_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

torch.set_num_threads(16)      # (the box's CPU share; torch's default there is 128 threads on a 16-CPU quota)

import test_densify_gpu as T  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad, t0 = [], time.time()
for seed in range(first, first + count):
    g = torch.Generator().manual_seed(5000 + seed)

    def u(a, b):
        return a + (b - a) * float(torch.rand((), generator=g))
    P = int(u(1, 30000)) if seed % 7 else int(u(1, 40))
    try:
        model, opt, accum, den = T._setup(P=P, seed=5100 + seed, with_moments=bool(seed % 3))
        params = {n: getattr(model, a).detach().cpu().clone() for n, a in zip(T.NAMES, T.ATTRS)}
        moments = {}
        for n, a in zip(T.NAMES, T.ATTRS):
            st = opt.state.get(getattr(model, a), {})
            moments[n] = (st["exp_avg"].cpu().clone(), st["exp_avg_sq"].cpu().clone()) if "exp_avg" in st else None
        has_m = all(v is not None for v in moments.values())
        extent, thr, min_op = u(3.0, 40.0), u(0.0001, 0.0012), u(0.001, 0.05)
        mss = None if seed % 2 else u(5.0, 45.0)
        ref_p, ref_m, info = T.DO.densify_and_prune(params, moments, accum.clone(), den.clone(),
                                                     model.max_radii2D.cpu().clone(), thr, min_op, extent, mss, model.percent_dense,
                                                     normal_samples=None)
        nk, nc, ns, src = model.densify_and_prune(thr, min_op, extent, mss, None, seed=seed, return_source=True)
        kind = info["kind"]
        assert (nk, nc, 2 * ns) == (int((kind == 0).sum()), int((kind == 1).sum()), int((kind == 2).sum())), "counts"
        assert torch.equal(src.cpu().long(), info["source"]), "row order"
        det, ch = kind != 2, kind == 2
        for n, a in zip(T.NAMES, T.ATTRS):
            got = getattr(model, a).detach().cpu()
            assert got.shape == ref_p[n].shape and torch.equal(got[det], ref_p[n][det]), n
            if has_m:
                st = opt.state[getattr(model, a)]
                assert torch.equal(st["exp_avg"].cpu(), ref_m[n][0]) and torch.equal(st["exp_avg_sq"].cpu(), ref_m[n][1]), n + " moments"
        for n in ("f_dc", "f_rest", "opacity", "rotation"):
            assert torch.equal(getattr(model, T.ATTRS[T.NAMES.index(n)]).detach().cpu()[ch], ref_p[n][ch]), n + " (split rows)"
        assert torch.allclose(model._scaling.detach().cpu()[ch], ref_p["scaling"][ch], atol=2e-6), "split scaling"
    except Exception as e:      # noqa: BLE001
        bad.append(seed)
        print(f"seed {seed} (P {P}): {type(e).__name__}: {str(e)[:200]}", flush=True)
    if (seed - first) % 25 == 24:
        print(f"... {seed - first + 1} cases, {len(bad)} failures, {time.time() - t0:.0f} s", flush=True)
print(f"densify sweep: seeds {first}..{first + count - 1}: {count - len(bad)} equal to the CPU restatement, {len(bad)} failed {bad}")
sys.exit(1 if bad else 0)
