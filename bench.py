#!/usr/bin/env python3
"""bench.py - BASELINE.json metric on the MI355X-native rasterizer path.

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run)

Workload (N=1 and N>1): BASELINE.json configs[2] - the configuration the metric is quoted on
("@1080p, 1M Gaussians"): 1,000,000 synthetic Gaussians, SH degree 3, 1920x1080, 100 synthetic views
(SURVEY.md Appendix C recipe, seeds 3000/3001).  A "step" is one training iteration per rank with the shape of
reference train.py:96-137,170-179: render() forward -> (1-l)L1 + l(1-SSIM) loss -> backward -> Adam step; with N
ranks every rank renders a different view and the 59 floats/Gaussian of gradients are all-reduced (RCCL) before
Adam.  `value` = view-iterations per second over the whole job (N*K / max-over-ranks time): weak scaling.

Extra keys on the same JSON line: `roofline` (dominant kernel, algorithmic bytes / measured HIP-event time),
`cpu_baseline` (the CPU oracle timed on a bounded sample on rank 0 at N=1), `fwd_mpix_per_s`, `kernels`.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-slam_amd"))
sys.path.insert(0, ROOT)
# (the pool's host driver supports dmabuf IPC only: without this RCCL's buffer sharing between the ranks of a node fails with
# hipIpcGetMemHandle "invalid argument"; the image exports it already - kept for environments that build their own)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3   # same guide, "Peak FP32 (vector)": what SURVEY.md 8(d)'s FLOP model of the compositing kernels is priced against
# SURVEY.md 8(d) FLOP model: forward 20 FLOP per (pixel, Gaussian) pair evaluated + 8 per pair blended; backward 70 per pair replayed
FWD_FLOP_PER_PAIR, FWD_FLOP_PER_BLEND, BWD_FLOP_PER_PAIR = 20, 8, 70
# FP32 vector issue peak: 256 CUs x 4 SIMD-32, a wave64 VALU instruction occupies its SIMD for 2 cycles (MI355X_MICROARCH.md
# "Wave scheduling" and the cycle-constants row `v_fma_f32 (wave64) 2 cyc (SIMD-32)`), 2.4 GHz -> 1.2288e12 wave-instr/s
# (= the 157.3 TFLOP/s fp32 vector peak / 128 flop per wave64 FMA).  profiles/r02_valu_microbench.txt measures it on the part.
VALU_WAVE_INSTR_PER_S = 256 * 4 * 2.4e9 / 2.0
_T0 = time.time()


_REAL_STDOUT_FD = None


def _claim_stdout():
    """The contract is ONE JSON line on stdout.  Native libraries write there too (librccl's version banner at communicator
    creation), so descriptor 1 is pointed at stderr for the life of the process and the line goes out through a duplicate of
    the original descriptor."""
    global _REAL_STDOUT_FD
    if _REAL_STDOUT_FD is None:
        sys.stdout.flush()
        _REAL_STDOUT_FD = os.dup(1)
        os.dup2(2, 1)


def _emit_json_line(line):
    sys.stdout.flush()
    fd = _REAL_STDOUT_FD if _REAL_STDOUT_FD is not None else 1
    data = (line + "\n").encode()
    while data:
        data = data[os.write(fd, data):]


def log(msg):
    """progress to stderr (stdout carries exactly one JSON line)"""
    print(f"[bench {time.time() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def host_threads():
    """CPU threads this process may really use (the GPU box gives a 16-CPU share of a bigger host)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("BENCH_CPU_THREADS", "16"))))


def build_scene(args, device, rank, world):
    from scene_utils import make_config, GaussianModel, make_gaussians
    from gaussian_renderer import render, PipelineParams
    raw, cams, cfg = make_config(args.config, device="cpu", P=args.gaussians, views=args.views,
                                 W=args.width, H=args.height, splat_scale=args.scale_factor)
    for c in cams:
        c.to(device)
    pipe = PipelineParams(antialiasing=bool(cfg.get("antialiasing", False)))
    bg = torch.zeros(3, device=device)
    # hidden "teacher": same recipe, different seed for colour/opacity perturbation -> ground-truth images
    teacher_raw = make_gaussians(cfg["P"], cfg["deg"], 1000 * args.config, scale_factor=0.25 * args.scale_factor)
    gen = torch.Generator().manual_seed(1000 * args.config + 7)
    teacher_raw.features_dc += 0.3 * torch.randn(teacher_raw.features_dc.shape, generator=gen)
    teacher_raw.xyz += 0.002 * torch.randn(teacher_raw.xyz.shape, generator=gen)
    teacher = GaussianModel.from_raw(teacher_raw.to(device), requires_grad=False)
    my_views = list(range(len(cams)))[rank::world]
    if not my_views:                       # fewer views than ranks (a one-view config under a launcher): share one
        my_views = [rank % len(cams)]
    log(f"scene generated: P={cfg['P']} views={len(cams)}")
    gts, depth_gts = {}, ({} if cfg.get("depth_grad") else None)

    def render_ground_truth():
        """Fills gts / depth_gts (the dicts the Trainer was given) with the teacher's renders of this rank's views.  Called as
        the LAST piece of setup, right in front of the warm-up steps: the model upload and the Trainer's construction in between
        would otherwise leave the device idle for ~0.1 s and the first timed steps would run at clocks still ramping up."""
        nonlocal teacher
        log(f"rendering {len(my_views)} teacher views")
        t_start = time.perf_counter()
        with torch.no_grad():
            for v in my_views:
                pkg = render(cams[v], teacher, pipe, bg)
                gts[v] = pkg["render"].clamp(0, 1).clone()
                if depth_gts is not None:
                    depth_gts[v] = pkg["depth"].clone()
            # A process that is the first to touch a cold GPU finds its clocks still ramping: the 100 teacher renders are 45 ms
            # of work, and the first timed steps of a 20-step run then come out 10 % slow (mean 1.45 ms against a median of 1.31,
            # profiles/README.md round 4).  Setup therefore keeps the device busy with more (discarded) teacher renders until
            # BENCH_PREWARM_MS (default 400) of wall time have passed here; it touches no state the timed steps use.
            prewarm = float(os.environ.get("BENCH_PREWARM_MS", "400")) * 1e-3
            i = 0
            while time.perf_counter() - t_start < prewarm:
                render(cams[my_views[i % len(my_views)]], teacher, pipe, bg)
                i += 1
                if i % 16 == 0:
                    torch.cuda.synchronize()
        teacher = None
        torch.cuda.synchronize()
        log("teacher views rendered")
    model = GaussianModel.from_raw(raw.to(device), requires_grad=True)
    if os.environ.get("BENCH_GT_FIRST") == "1":      # (A/B switch: the old order)
        render_ground_truth()
        render_ground_truth = lambda: None           # noqa: E731
    return model, cams, gts, depth_gts, my_views, pipe, bg, cfg, render, render_ground_truth


def _pmc(kernel, cfg_key):
    """Per-launch PMC figures of `kernel` from the committed rocprofv3 passes (profiles/pmc_latest.json, made by
    tools/pmc_summary.py: FETCH_SIZE x 1024 x 2 [gfx950 correction] + WRITE_SIZE x 1024, separate passes) - only when they
    were collected on THIS workload (`workload` key of the file), otherwise None: counters of one config say nothing about
    another."""
    try:
        doc = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
    except Exception:
        return None
    if "workloads" in doc:
        doc = doc["workloads"].get(cfg_key)
        if doc is None:
            return None
    elif doc.get("workload") != cfg_key:
        return None
    # profile label -> kernel symbol(s): the backward has two forms, the folded optimizer step is a template instance
    want = {"render_bwd": ("k_render_bwd_tile", "k_render_bwd"), "preprocess_bwd_adam": ("k_preprocess_bwd",),
            "adam_dense": ("k_adam",), "adam_sparse": ("k_adam",), "radix_hist": ("k_radix_hist_all",)}.get(kernel, ("k_" + kernel,))
    for sym in want:
        for name, v in doc["kernels"].items():
            if name.split("<")[0] == sym:
                return v
    return None


def pmc_traffic(kernel, cfg_key):
    v = _pmc(kernel, cfg_key)
    return int(v["traffic_bytes"]) if v and "traffic_bytes" in v else None


def pmc_valu(kernel, avg_ms, cfg_key):
    """FP32 VALU issue bound of `kernel` from SQ_INSTS_VALU (wave-level instructions per launch): no schedule finishes
    faster than insts / VALU_WAVE_INSTR_PER_S.  `issue_frac` = that bound / the measured duration (<= 1 by construction)."""
    v = _pmc(kernel, cfg_key)
    if not v or "valu_insts" not in v:
        return None
    bound_ms = v["valu_insts"] / VALU_WAVE_INSTR_PER_S * 1e3
    return {"insts_per_launch": int(v["valu_insts"]), "cycles_per_wave_instr": 2, "issue_bound_ms": round(bound_ms, 4),
            "issue_frac": round(bound_ms / avg_ms, 3) if avg_ms else None}


def kernel_table(prof, R, N, P, M):
    """Algorithmic HBM bytes per launch (SURVEY.md 8(d) per-unit figures x units per launch; DESIGN.md 'Kernels')."""
    b_in = 44 + 12 * M
    alg = {
        "preprocess_fwd": P * (b_in + 75),
        "render_fwd": 44 * R + 24 * N,
        "render_bwd": 84 * R + 24 * N,
        "preprocess_bwd": P * (2 * b_in + 79) + 44 * R,
        # optimizer folded in: no gradient stores (b_in), parameters read once (already counted), both moments read, parameters
        # and both moments written: + 5 x (11 + 3 M) floats per Gaussian
        "preprocess_bwd_adam": P * (b_in + 79) + 44 * R + 20 * (11 + 3 * M) * P,
        "emit_instances": 44 * P + 8 * R,          # order, offsets, 32-B binning record, slot_start; tile id + Gaussian id out
        "finalize_bins": 4 * R,                    # sorted tile ids in, 8 B per tile out
        "adam_dense": 28 * (11 + 3 * M) * P,       # param, grad, two moments in; param, two moments out
        "sum_tiles": 4 * P,
        "densify_stats": 28 * P,
    }
    if "shade" in prof:
        # colour pass in its own kernel (side stream): the projection kernel no longer reads the SH rows or writes colours
        alg["preprocess_fwd"] = P * (44 + 75)
        alg["shade"] = P * (12 * M + 12 + 16 + 1)          # SH row + mean in; colour part of the record + clamp flags out
    if "tile_depth_sort" in prof:
        # tile-local binning form: list entry + its depth key in, (entry, emission slot) re-read through the permutation and
        # written back; the tile sort's passes all run on the R instances
        alg["tile_depth_sort"] = 28 * R
        alg["radix_hist"] = 4 * R
        alg["emit_instances"] = 40 * P + 8 * R
    out = {}
    for name, (ms, calls) in prof.items():
        if calls == 0:
            continue
        avg = ms / calls
        e = {"avg_ms": round(avg, 4), "calls": int(calls)}
        if name in alg:
            e["alg_bytes"] = int(alg[name])
            e["gbps"] = round(alg[name] / (avg * 1e-3) / 1e9, 1)
        out[name] = e
    return out


def cpu_baseline(args, model, cam, gt, bg, cfg, gpu_image):
    """CPU oracle (pure-PyTorch restatement, `kind: port`) on a bounded sample of the SAME view:
    full preprocess + binning, forward+backward compositing of n1 and n2 tiles; per-tile slope extrapolated to
    all tiles.  Also returns PSNR(GPU image, oracle) on the sampled tiles."""
    from oracle import gs_oracle as O

    def oracle_settings(cam, deg, bg_, antialiasing):
        return O.OracleSettings(int(cam.image_height), int(cam.image_width), math.tan(cam.FoVx * 0.5),
                                math.tan(cam.FoVy * 0.5), bg_, 1.0, cam.world_view_transform.cpu(),
                                cam.full_proj_transform.cpu(), deg, cam.camera_center.cpu(), False, False, antialiasing)
    torch.set_num_threads(host_threads())
    W, H = cfg["W"], cfg["H"]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    ntiles = gx * gy
    gen = torch.Generator().manual_seed(5)
    perm = torch.randperm(ntiles, generator=gen).tolist()
    n1, n2 = (args.cpu_tiles, 3 * args.cpu_tiles)
    n2 = min(n2, ntiles)
    n1 = min(n1, n2)
    times = []
    psnr_db = None
    for n in (n1, n2):
        tiles = sorted(perm[:n])
        leaves = [p.detach().cpu().clone().requires_grad_(True) for p in model.parameters()]
        xyz, fdc, frest, opac, scal, rot = leaves
        t0 = time.perf_counter()
        s = oracle_settings(cam, cfg["deg"], bg.cpu(), antialiasing=bool(cfg.get("antialiasing", False)))
        m2d = torch.zeros(xyz.shape[0], 3, requires_grad=True)
        color, radii, invd = O.rasterize(xyz, m2d, torch.sigmoid(opac), s, shs=torch.cat((fdc, frest), 1),
                                         scales=torch.exp(scal), rotations=torch.nn.functional.normalize(rot),
                                         tiles=tiles)
        mask = torch.zeros(1, H, W)
        for t in tiles:
            ty, tx = divmod(t, gx)
            mask[:, ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16] = 1
        loss = (torch.abs(color - gt.cpu()) * mask).sum() / (3 * H * W)
        loss.backward()
        times.append(time.perf_counter() - t0)
        log(f"cpu oracle: {n} tiles fwd+bwd in {times[-1]:.1f}s")
        if n == n2:
            sel = mask.bool().expand(3, H, W)
            mse = ((color.detach() - gpu_image.cpu())[sel] ** 2).mean().item()
            psnr_db = 99.0 if mse == 0 else 10 * math.log10(1.0 / mse)
    if n1 == n2 == ntiles:
        # the whole image fits the sample (BASELINE configs[0]): nothing to extrapolate, the second (warm) run is the figure
        est = times[1]
        unit = "view-iterations/s (fwd+bwd, measured on the full image)"
        sample = (f"view 0 of the bench scene, all {ntiles} tiles, forward+backward, run twice "
                  f"({times[0]:.2f}s cold, {times[1]:.2f}s warm)")
    else:
        per_tile = (times[1] - times[0]) / max(1, (n2 - n1))
        fixed = max(0.0, times[0] - per_tile * n1)
        est = fixed + per_tile * ntiles
        unit = "view-iterations/s (fwd+bwd, extrapolated)"
        sample = (f"view 0 of the bench scene: full preprocess+binning+their backward ({fixed:.1f}s) plus "
                  f"{n1} and {n2} of {ntiles} tiles composited fwd+bwd ({per_tile * 1e3:.1f} ms/tile slope), "
                  f"scaled to {ntiles} tiles; CPU work measured {times[0] + times[1]:.1f}s")
    return {
        "value": round(1.0 / est, 5), "unit": unit,
        "cores": torch.get_num_threads(), "kind": "port",
        "sample": sample,
        "measured_s": round(times[0] + times[1], 2),
        "psnr_gpu_vs_oracle_db_on_sample": None if psnr_db is None else round(psnr_db, 2),
    }


def pair_count(model, cam, bg, cfg, dgr):
    rs = dgr.GaussianRasterizationSettings(
        int(cam.image_height), int(cam.image_width), math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg, 1.0,
        cam.world_view_transform, cam.full_proj_transform, cfg["deg"], cam.camera_center, False, False,
        bool(cfg.get("antialiasing", False)))
    sc, ro, op = model.get_raw_geometry()
    return dgr.pair_evaluations(rs, model.get_xyz, op, shs=model.get_features_rest, dc=model.get_features_dc, scales=sc,
                                rotations=ro, raw_activations=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--gaussians", type=int, default=None)
    ap.add_argument("--views", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--scale-factor", type=float, default=1.0,
                    help="multiplier on the recipe's splat size (SURVEY App. C).  1.0 = the BASELINE configs; 2.0 = the same scene "
                         "with twice the splat size: ~3.5x the tile instances per Gaussian, the compositing kernels dominate as on "
                         "a real capture (profiles/r04_bench_c3_heavy.json).  The JSON line names it in config.workload")
    ap.add_argument("--cpu-tiles", type=int, default=256)
    ap.add_argument("--optimizer", default="hip_fused", choices=["hip", "hip_fused", "hip_sparse", "hip_sparse_fused", "torch"],
                    help="hip: one-launch Adam kernel (torch.optim.Adam semantics, the reference's default optimizer); "
                         "hip_sparse: SparseGaussianAdam (reference train.py:173-176); *_fused: the same update folded into "
                         "the rasterizer's backward (gsr_backward_adam), bit-identical results")
    ap.add_argument("--loss", default="hip", choices=["hip"])
    ap.add_argument("--concat-sh", action="store_true",
                    help="pass torch.cat(dc, rest) as shs (separate_sh=False); default mirrors reference train.py:106 with "
                         "SparseGaussianAdam importable: separate_sh=True")
    ap.add_argument("--densify", action="store_true",
                    help="run the reference densify/prune schedule (train.py:155-168) inside the timed loop (config 5)")
    ap.add_argument("--densify-from", type=int, default=500)
    ap.add_argument("--densify-grad-threshold", type=float, default=0.0002,
                    help="densify_grad_threshold (reference arguments/__init__.py:90: 0.0002); the synthetic C5 scene needs a "
                         "lower one to grow the way a real capture does")
    ap.add_argument("--densify-max", type=int, default=None,
                    help="stop densifying once the model has this many Gaussians (C5: 500000)")
    ap.add_argument("--views-per-step", type=int, default=1,
                    help="views per rank per optimizer step (gradient accumulation; default 1 = the reference's batch-1 step). "
                         "Amortises the N>1 gradient exchange and Adam over k views; value still counts view-iterations")
    ap.add_argument("--exchange", default="auto", choices=["auto", "allreduce", "visible_rows", "sharded", "sh_rank1"],
                    help="N>1 gradient exchange.  allreduce: one RCCL all-reduce per leaf tensor, 59 floats per Gaussian (the "
                         "north-star schedule as written); sh_rank1: the SAME mean gradients with 2.6x fewer bytes on the links - "
                         "the 11 geometry floats all-reduced, the 48 SH floats rebuilt on every rank from an all-gather of "
                         "dL/df_dc + camera centres (a view's SH gradient is rank one per Gaussian); visible_rows: the "
                         "all-reduce restricted to the rows some rank saw; sharded: reduce-scatter -> Adam on a 1/N row shard "
                         "-> all-gather.  auto (default): sh_rank1 with one view per rank per step, else allreduce (DESIGN.md 5)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N>1: exchange all gradients on the main stream (default overlaps the SH exchange + Adam with the next "
                         "step's geometry stages; same results, DESIGN.md 5)")
    ap.add_argument("--overlap", action="store_true",
                    help="force the side-stream SH update on at N=1 too (default: on only for N>1, where it hides the exchange)")
    ap.add_argument("--hi-prio", action="store_true",
                    help="run the training loop on a high-priority stream (the side stream of --overlap stays at normal "
                         "priority, so the small binning kernels are dispatched ahead of the bandwidth-bound SH update)")
    ap.add_argument("--split-rows", action="store_true",
                    help="hip_fused: update the rows without tile instances on a side stream beside the compositing backward")
    ap.add_argument("--graph", action="store_true",
                    help="N=1: the training step as ONE HIP graph launch (Trainer.enable_graph_replay: captured once per model "
                         "size, replayed; same results bit for bit).  Pays where the step is launch-bound (configs 1, 5)")
    ap.add_argument("--cull", action="store_true",
                    help="tile lists truncated by depth (Trainer.enable_tile_cull; implies --forward-mode async unless one is given): "
                         "every view remembers the depth its tiles saturated at, its next render emits only what lies in front; a frame "
                         "whose truncation was too tight is a no-op on the device and is run again - same parameters bit for bit")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--forward-mode", default=None, choices=["exact", "async", "sync"],
                    help="rasterizer forward: exact (default: the frame is enqueued whole for a capacity estimate, its count is "
                         "verified before the call returns and phase 2 repeated if the estimate did not hold - every frame "
                         "exact), async (no wait at all: a truncated frame's backward is a no-op and the Trainer runs the view "
                         "again) or sync (the published blocking read-back in the middle of the forward)")
    args = ap.parse_args()
    if args.config == 5 and not args.densify and os.environ.get("BENCH_C5_STATIC") != "1":
        # BASELINE configs[4] "SLAM-style incremental: 50k -> 500k growing Gaussians, per-frame 720p fwd+bwd, densify/prune every
        # 100 iters": the reference's schedule (train.py:155-168, arguments/__init__.py:84-90) from iteration 100; its 0.0002
        # gradient threshold stops the SYNTHETIC scene near 100 k, so the preset uses 0.00002 and stops at 500 k
        args.densify = True
        if "--densify-from" not in sys.argv:
            args.densify_from = 100
        if "--densify-grad-threshold" not in sys.argv:
            args.densify_grad_threshold = 0.00002
        if args.densify_max is None:
            args.densify_max = 500000
        if "--steps" not in sys.argv:
            args.steps = 2500

    # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (torch.distributed.run, one process per
    # GPU) BEFORE anything in this process touches the GPU, relay the child's stdout / exit code.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        if torch.cuda.device_count() < args.gpus and not os.environ.get("BENCH_SHARE_GPU"):   # (does not initialise the GPU)
            print(f"bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} visible", file=sys.stderr)
            sys.exit(2)
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        log("no launcher detected: " + " ".join(cmd))
        sys.exit(subprocess.call(cmd))

    _claim_stdout()
    if args.exchange == "auto":
        args.exchange = "sh_rank1" if max(1, args.views_per_step) == 1 and args.optimizer in ("hip", "hip_fused", "torch") \
            else "allreduce"
    torch.set_num_threads(host_threads())
    from scene_utils import init_from_env, Trainer, exchange_bytes_per_gaussian
    # BENCH_BACKEND=gloo + BENCH_SHARE_GPU=1 rehearse the N>1 path with several ranks on ONE card (no RCCL between them)
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    rank, world, local = init_from_env(backend)
    # BENCH_SINGLE_RANK_GROUP=1 (with --gpus 1): a process group of ONE rank, and the Trainer runs the whole N > 1 schedule on it
    # (every collective through RCCL, side stream, exchange kernels): the rehearsal of that path on a one-GPU box.  Not a
    # scaling number: the line it prints says n_gpus 1 and "single_rank_group": true.
    single_rank_group = world == 1 and os.environ.get("BENCH_SINGLE_RANK_GROUP") == "1"
    if single_rank_group:
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        torch.cuda.set_device(0)
        dist.init_process_group(backend=backend, init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    if os.environ.get("BENCH_SHARE_GPU"):
        local = 0
    log(f"rank {os.environ.get('RANK', '0')} start; host threads {host_threads()} (cpu_count {os.cpu_count()})")
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank number as "
                  f"{args.gpus} GPUs", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    import diff_gaussian_rasterization as dgr
    if args.cull and not args.forward_mode:
        args.forward_mode = "async"
    if args.forward_mode:
        dgr.set_forward_mode(args.forward_mode)
    model, cams, gts, depth_gts, my_views, pipe, bg, cfg, render, render_ground_truth = build_scene(args, device, rank, world)
    trainer = Trainer(model, cams, gts, render, pipe, bg, world=world, rank=rank, optimizer=args.optimizer,
                      loss=args.loss, separate_sh=not args.concat_sh, depth_targets=depth_gts,
                      depth_weight=1.0 if depth_gts is not None else 0.0,
                      overlap_comm=False if args.no_overlap else (True if args.overlap else None),
                      exchange=args.exchange, single_rank_group=single_rank_group)
    trainer.split_rows = bool(args.split_rows)
    if args.cull:
        trainer.enable_tile_cull()
    if args.graph:
        if world > 1 or single_rank_group:
            raise SystemExit("--graph: one rank only")
        trainer.enable_graph_replay()
    if args.densify:
        # cameras_extent of the reference = 1.1 x radius of the camera centres (scene/dataset_readers.py getNerfppNorm)
        trainer.enable_densification(extent=1.1 * 4.0, from_iter=args.densify_from,
                                     grad_threshold=args.densify_grad_threshold, max_gaussians=args.densify_max)
    P = cfg["P"]
    M = (cfg["deg"] + 1) ** 2
    W, H = cfg["W"], cfg["H"]

    def view_at(i):
        return my_views[i % len(my_views)]

    k = max(1, args.views_per_step)

    def views_of_step(i):
        return view_at(i) if k == 1 else [view_at(i * k + j) for j in range(k)]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.hi_prio:
        hi = torch.cuda.Stream(device=device, priority=-1)
        hi.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(hi)
    render_ground_truth()
    # (a cyclic-GC pass over everything setup allocated - 100 cameras, ground-truth tensors, the scene - costs milliseconds when it
    # happens to fall into a 26 ms timed region: collect now and exempt what exists from further passes)
    import gc
    gc.collect()
    gc.freeze()
    for i in range(args.warmup):
        trainer.step(views_of_step(i))
    trainer.finish()       # an SH update handed to "the next forward" belongs to the step that produced it: flush it here ...
    barrier()
    log("warmup done")
    alloc0 = torch.cuda.memory_stats(device).get("num_device_alloc", 0)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    stats_before = dict(dgr.call_stats(device, wait=False))
    host_ms = []
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        h0 = time.perf_counter()
        trainer.step(views_of_step(args.warmup + i))
        marks[i + 1].record()          # per-step GPU timeline (the host runs ahead of the device: no wait in the loop)
        host_ms.append((time.perf_counter() - h0) * 1e3)
    trainer.finish()       # ... and here, so the timed region holds exactly K complete optimizer steps
    barrier()
    elapsed = time.perf_counter() - t0
    step_seq = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    step_ms = sorted(step_seq)
    host_ms.sort()

    def pct(v, q):
        return round(v[min(len(v) - 1, int(q * len(v)))], 4)
    timed_stats = dgr.call_stats(device)
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    log(f"timed region done: {elapsed / args.steps * 1e3:.2f} ms/step; Gaussians now {model.get_xyz.shape[0]}")
    result = {
        "step_ms_sequence": [round(x, 3) for x in step_seq],
        **({"graph_replay": dict(trainer.graph_stats)} if args.graph else {}),
        "metric": "train_iters_per_sec", "value": round(world * k * args.steps / elapsed, 3),
        "unit": "view-iterations/s (render fwd + L1/DSSIM loss + bwd + Adam)", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        # distribution of the K timed steps on the GPU timeline (HIP events between steps) and of the host's enqueue time
        "ms_per_step_median": pct(step_ms, 0.5), "ms_per_step_p10": pct(step_ms, 0.1), "ms_per_step_p90": pct(step_ms, 0.9),
        "host_ms_per_step_median": pct(host_ms, 0.5), "host_ms_per_step_p90": pct(host_ms, 0.9),
        "device_allocs_in_timed_region": int(torch.cuda.memory_stats(device).get("num_device_alloc", 0) - alloc0),
        # frames of the whole run (warm-up included) by forward form; tile_local_frames_timed of the K*views timed frames
        "forward_mode": {"mode": dgr.forward_mode(), "exact_frames": timed_stats["exact_frames"],
                         "async_frames": timed_stats["async_frames"], "sync_frames": timed_stats["sync_frames"],
                         "rerendered_frames": timed_stats["rerendered_frames"],
                         "overflow_frames": timed_stats["overflow_frames"], "rerun_views": trainer.rerun_views,
                         "tile_cull": bool(args.cull), "culled_frames": timed_stats.get("culled_frames", 0),
                         "cull_miss_frames": timed_stats.get("cull_miss_frames", 0),
                         "tile_local_frames": timed_stats.get("tile_local_frames", 0),
                         "tile_local_frames_timed": timed_stats.get("tile_local_frames", 0)
                                                    - stats_before.get("tile_local_frames", 0)},
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": (f"NOT a BASELINE config - configs[{args.config - 1}] with {args.scale_factor}x the recipe's splat size: "
                                if args.scale_factor != 1.0 else f"BASELINE configs[{args.config - 1}]: ") +
                               f"{P} Gaussians, SH degree {cfg['deg']}, "
                               f"{W}x{H}, {len(cams)} views, one view per rank per step, mean of the ranks' "
                               f"{59 if cfg['deg'] == 3 else 11 + 3 * M}-float/Gaussian gradients over RCCL when N>1 "
                               f"(exchange: {trainer.exchange})",
                   "gaussians": P, "gaussians_final": int(model.get_xyz.shape[0]), "densify": bool(args.densify),
                   "densify_grad_threshold": args.densify_grad_threshold if args.densify else None,
                   "densify_max": args.densify_max if args.densify else None,
                   "sh_degree": cfg["deg"], "width": W, "height": H, "views": len(cams),
                   "antialiasing": bool(cfg.get("antialiasing", False)), "parallelism": f"view-dp{world}", 
                   **({"single_rank_group": True} if single_rank_group else {}),
                   "views_per_rank_per_step": k, "overlap_comm": bool(trainer.overlap_comm), "exchange": trainer.exchange,
                   "exchange_bytes_per_gaussian_received": round(exchange_bytes_per_gaussian(trainer.exchange, world, M), 1)
                                                            if world > 1 else 0,
                   "loss": "L1 + 0.2 DSSIM (" + ("HIP fused SSIM" if args.loss == "hip" else "torch conv2d SSIM") + ")",
                   "optimizer": {"hip": "Adam, one-launch HIP kernel (torch.optim.Adam semantics)",
                                 "hip_fused": "Adam (torch.optim.Adam semantics) folded into the rasterizer backward",
                                 "hip_sparse": "SparseGaussianAdam (HIP)",
                                 "hip_sparse_fused": "SparseGaussianAdam folded into the rasterizer backward",
                                 "torch": "torch.optim.Adam"}[args.optimizer if not trainer.distributed else
                                                              args.optimizer.replace("_fused", "")] +
                                (" (SH groups: step applied by the sh_rank1 rebuilding kernel)"
                                 if trainer.distributed and trainer.exchange == "sh_rank1" and trainer.rank1_fuse_adam
                                 and args.optimizer in ("hip", "hip_fused") and k == 1 else "")},
    }

    # ---- untimed extras (rank 0 reports) ----
    from diff_gaussian_rasterization import _C
    lib = _C.lib()
    # forward-only throughput (reference render.py:37-49 path: no_grad)
    with torch.no_grad():
        for _ in range(3):
            render(cams[view_at(0)], model, pipe, bg, separate_sh=not args.concat_sh)
        torch.cuda.synchronize()
        nf = max(5, min(30, args.steps))
        t0 = time.perf_counter()
        for i in range(nf):
            render(cams[view_at(i)], model, pipe, bg, separate_sh=not args.concat_sh)
        torch.cuda.synchronize()
        fwd_s = (time.perf_counter() - t0) / nf
    result["fwd_mpix_per_s"] = round(W * H / fwd_s / 1e6, 1)
    result["fwd_ms"] = round(fwd_s * 1e3, 3)

    if not args.no_kernel_profile:
        lib.gsr_profile_enable(1)
        lib.gsr_profile_reset()
        nprof = max(3, min(10, args.steps))
        Rs = []
        for i in range(nprof):
            trainer.step(view_at(i))
            Rs.append(dgr.call_stats(device)["num_rendered"])   # measured R of each profiled view (waits for the count)
        trainer.finish()
        torch.cuda.synchronize()
        prof = _C.profile_read()
        lib.gsr_profile_enable(0)
        R = sum(Rs) / len(Rs)
        kt = kernel_table(prof, R, W * H, int(model.get_xyz.shape[0]), M)
        result["num_rendered_avg"] = int(R)
        result["kernels"] = kt
        result["kernel_ms_sum"] = round(sum(v["avg_ms"] * v["calls"] for v in kt.values()) / nprof, 4)
        # pair evaluations of the two compositing kernels (SURVEY.md 8(d) FLOP model), view 0
        try:
            pe = pair_count(model, cams[view_at(0)], bg, cfg, dgr)
            for kname, key in (("render_fwd", "fwd_pairs"), ("render_bwd", "bwd_pairs")):
                if kname in kt:
                    kt[kname]["pairs"] = pe[key]
                    kt[kname]["gpairs_per_s"] = round(pe[key] / (kt[kname]["avg_ms"] * 1e-3) / 1e9, 1)
            result["pair_evaluations"] = pe
        except Exception as e:   # an extra; never lose the headline to it
            result["pair_evaluations"] = {"error": repr(e)}
        # FLOP model of the two compositing kernels (pairs are counted on view 0, durations are the profiled average)
        pe = result.get("pair_evaluations", {})
        flops = {}
        if "fwd_pairs" in pe:
            flops["render_fwd"] = FWD_FLOP_PER_PAIR * pe["fwd_pairs"] + FWD_FLOP_PER_BLEND * pe.get("fwd_blended", 0)
            flops["render_bwd"] = BWD_FLOP_PER_PAIR * pe["bwd_pairs"]
        cfg_key = f"c{args.config}:{P}:{W}x{H}" + (f":s{args.scale_factor}" if args.scale_factor != 1.0 else "")

        def roof_of(name):
            d = kt[name]
            traffic = pmc_traffic(name, cfg_key)
            hbm = {"achieved": d["gbps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(d["gbps"] / HBM_PEAK_GBS, 5),
                   "alg_bytes": d["alg_bytes"], "traffic": traffic,
                   "traffic_ratio": round(traffic / d["alg_bytes"], 3) if traffic else None}
            valu = pmc_valu(name, d["avg_ms"], cfg_key)      # issue utilisation (SQ_INSTS_VALU of the committed PMC pass): NOT a roofline fraction
            roof = {"kernel": name, "avg_ms": d["avg_ms"]}
            if name in flops:
                # compositing kernels: records staged through LDS, few bytes moved - priced by SURVEY 8(d)'s FLOP model against
                # the FP32 vector peak; the HBM view and the VALU issue utilisation ride along
                tf = flops[name] / (d["avg_ms"] * 1e-3) / 1e12
                roof.update({"bound": "valu", "achieved": round(tf, 2), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(tf / FP32_PEAK_TFLOPS, 4), "flop_model": flops[name], "traffic": traffic,
                             "traffic_ratio": hbm["traffic_ratio"], "hbm": hbm, "valu_issue": valu})
            else:
                roof.update({"bound": "hbm", **hbm})
            return roof
        named = {k: v for k, v in kt.items() if "alg_bytes" in v}
        if named:
            dom = max(named, key=lambda k: named[k]["avg_ms"] * named[k]["calls"])
            result["roofline"] = roof_of(dom)
            # the other kernels above 5 % of the step, same pricing, for the record
            total = sum(v["avg_ms"] * v["calls"] for v in kt.values())
            result["rooflines"] = {k: {kk: vv for kk, vv in roof_of(k).items() if kk in ("bound", "achieved", "peak", "unit", "frac", "traffic_ratio")}
                                   for k in named if k != dom and named[k]["avg_ms"] * named[k]["calls"] > 0.05 * total}

    log("extras done; cpu baseline next")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        v0 = view_at(0)
        with torch.no_grad():
            gpu_img = render(cams[v0], model, pipe, bg)["render"]
        try:
            result["cpu_baseline"] = cpu_baseline(args, model, cams[v0], gts[v0], bg, cfg, gpu_img)
        except Exception as e:  # the baseline is a reported extra; never lose the headline number to it
            result["cpu_baseline"] = {"value": None, "error": repr(e)}

    if rank == 0:
        # ONE line on the real stdout (everything else that writes to descriptor 1 - RCCL prints a version banner there - was sent
        # to stderr at start-up), written and flushed before the process group is touched again: a teardown that aborts would
        # otherwise take a block-buffered line with it
        _emit_json_line(json.dumps(result))
    if world > 1 or single_rank_group:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception as e:      # the number is out; a failing teardown must not turn the run into an error
            log(f"process-group teardown: {e!r}")


if __name__ == "__main__":
    main()
