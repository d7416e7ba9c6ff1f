"""MI355X-native drop-in for the `diff_gaussian_rasterization` module the reference imports at
`gaussian_renderer/__init__.py:14`:

    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

Same Python surface: `GaussianRasterizationSettings` (13 fields, order of reference
`gaussian_renderer/__init__.py:36-50`), `GaussianRasterizer(raster_settings)` whose call returns
`(color[3,H,W], radii[P] int32, invdepth[1,H,W])` (reference :90-109), `markVisible`, the `means2D.grad`
side channel (reference :26-30; consumed at `scene/gaussian_model.py:431-433`).  All arithmetic runs in
hand-written HIP kernels behind the C ABI of `include/gsr.h`; there is no PyTorch/CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C
from . import _workspace as _ws
from ._workspace import set_forward_mode, forward_mode  # noqa: F401


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    antialiasing: bool = False


class BackwardFold:
    """Per-call request - and receipt - for work the rasterizer's backward can do in its last kernel.  Pass it to THE forward whose
    backward is meant (`GaussianRasterizer.forward(..., fold=...)`, `render(..., fold=...)`); it travels on that call's autograd
    ctx, so any other render / backward in between (a viewer frame, an evaluation view, another device or thread) neither sees
    nor consumes it (no global state, re-entrant per device).

    optimizer      a FusedAdam / SparseGaussianAdam over the model's six parameter groups (named xyz, f_dc, f_rest, opacity,
                   scaling, rotation as in reference scene/gaussian_model.py:160-168): the backward of a call that received
                   exactly those parameters (raw-parameter call form, dc / rest separate) applies the Adam update itself
                   (gsr_backward_adam) and returns no gradient for them.  `optimizer_taken` says whether it did; if not, the
                   gradients are in `.grad` as usual and the caller runs optimizer.step().
    split_rows     (dense Adam only) the rows WITHOUT tile instances in that forward (exact zero gradient) are updated by
                   gsr_adam_step_culled_rows on a side stream while the compositing backward - bound by VALU issue, the HBM
                   idle - runs on the caller's stream; bit-identical to the unsplit update.
    stats          (xyz_gradient_accum [P,1], denom [P,1], max_radii2D [P]): the backward also performs this view's
                   `add_densification_stats` (reference scene/gaussian_model.py:431-433) and the `max_radii2D` update
                   (train.py:159) - same arithmetic as gsr_densification_stats, no extra pass, no extra launch (`stats_taken`).
    skip_sh_rest   (view-sharded exchange "sh_rank1", scene_utils/parallel.py) a call that got `dc` and `shs` separately forms
                   dL/ddc only and returns None for `shs`: the ranks exchange dL/ddc and rebuild the other coefficients' mean
                   gradient from it, so this rank's own 180 B per Gaussian are not written at all (`sh_rest_skipped`)."""

    __slots__ = ("optimizer", "split_rows", "stats", "skip_sh_rest", "optimizer_taken", "stats_taken", "sh_rest_skipped")

    def __init__(self, optimizer=None, split_rows=False, stats=None, skip_sh_rest=False):
        self.optimizer = optimizer
        self.split_rows = bool(split_rows) and optimizer is not None
        self.stats = None if stats is None or stats[0] is None else tuple(stats)
        self.skip_sh_rest = bool(skip_sh_rest)
        self.optimizer_taken = self.stats_taken = self.sh_rest_skipped = False


_side_streams = {}         # device index -> side stream of the split optimizer update (a cache, not a hand-off)
GRAD_ARENA_ALIGN = 64      # floats: every gradient tensor of the arena starts on a 256-byte boundary
GRAD_ARENA_SPARE = 3       # floats left free behind EVERY tensor of the arena (so the layout is a function of the sizes alone:
#                            scene_utils.parallel.canonical_offsets predicts it on every rank without looking at addresses)
_SPARE = "_gsr_spare_floats"   # attribute of an arena's STORAGE: {(element offset, numel): spare floats behind} - see dc_grad_tail_row


def _grad_arena(dev, parts):
    """parts: ((shape | None, spare floats wanted behind it), ...) -> one tensor per part (None where the shape is None), all
    views of ONE float32 allocation, in order, each starting on a GRAD_ARENA_ALIGN boundary, each followed by at least
    GRAD_ARENA_SPARE free floats."""
    offs, total = [], 0
    for shape, spare in parts:
        if shape is None:
            offs.append(None)
            continue
        n = 1
        for d in shape:
            n *= int(d)
        offs.append((total, n))
        total += -(-(n + max(int(spare), GRAD_ARENA_SPARE)) // GRAD_ARENA_ALIGN) * GRAD_ARENA_ALIGN
    flat = torch.empty(total, dtype=torch.float32, device=dev)
    # what is spare behind which tensor is a property of THIS allocation: it is recorded on the storage object, which stays the
    # same object when autograd moves the tensor into `.grad` and is a different one for any copy / accumulated sum of it
    # (plain numbers only: a tensor stored on its own storage would be a reference cycle no collector can see)
    spares = {o: max(int(spare), GRAD_ARENA_SPARE) for o, (shape, spare) in zip(offs, parts) if o is not None}
    if spares:
        setattr(flat.untyped_storage(), _SPARE, spares)
    return [None if o is None else flat[o[0]:o[0] + o[1]].view(shape) for o, (shape, _) in zip(offs, parts)]


def dc_grad_tail_row(g):
    """`g`: a dc gradient ([P, 1, 3]).  If it still IS the tensor a rasterizer backward returned (its storage is that backward's
    gradient arena - not a copy, not a sum accumulated into another buffer), the 3 floats behind it are spare: returns the
    [P + 1, 3] tensor over both (the sh_rank1 exchange writes the camera centre into the last row instead of concatenating 12 B
    per Gaussian), else None.  No process-wide "most recent backward": the knowledge rides on the allocation itself."""
    if g is None or not g.is_contiguous() or g.dtype != torch.float32:
        return None
    spares = getattr(g.untyped_storage(), _SPARE, None)
    if spares is None or spares.get((g.storage_offset(), g.numel()), 0) < 3 or g.numel() % 3:
        return None
    return torch.empty(0, dtype=g.dtype, device=g.device).set_(g.untyped_storage(), g.storage_offset(), (g.numel() // 3 + 1, 3), (3, 1))


_GROUP_ORDER = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")


def _fused_adam_struct(opt, tensors, advance=True, rows=None):
    """gsr_fused_adam for optimizer `opt` if its six groups are exactly `tensors` (means3D, dc, sh, opacities, scales,
    rotations as saved by the forward), else None.  advance=False: the step numbers this update WILL carry, without
    counting it yet (the early culled-rows half of a split update)."""
    groups = {g.get("name"): g for g in opt.param_groups}
    if any(n not in groups for n in _GROUP_ORDER):
        return None
    ps = []
    for n, t in zip(_GROUP_ORDER, tensors):
        p = groups[n]["params"][0]
        if t is None:
            if p.numel() != 0:
                return None
        elif p.data_ptr() != t.data_ptr() or p.numel() != t.numel() or not p.is_contiguous() or p.dtype != torch.float32:
            return None
        ps.append(p)
    fa = _C.gsr_fused_adam()
    sparse = isinstance(opt, SparseGaussianAdam)
    keep = []
    for i, (n, p) in enumerate(zip(_GROUP_ORDER, ps)):
        st = opt.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        if not sparse and advance:
            st["step"] += 1
        fa.exp_avg[i] = st["exp_avg"].data_ptr() if p.numel() else None
        fa.exp_avg_sq[i] = st["exp_avg_sq"].data_ptr() if p.numel() else None
        fa.lr[i] = float(groups[n]["lr"])
        fa.step[i] = (int(st["step"]) + (0 if advance else 1)) if not sparse else 1
        keep.append(st)
    g0 = opt.param_groups[0]
    b1, b2 = (0.9, 0.999) if sparse else g0["betas"]
    fa.beta1, fa.beta2, fa.eps = float(b1), float(b2), float(g0["eps"])
    fa.sparse = 1 if sparse else (2 if rows == "with_instances" else 0)
    dyn = getattr(opt, "_gsr_dynamic", None)
    fa.dynamic = dyn.data_ptr() if dyn is not None else None
    return fa, keep


def enable_dynamic_hyperparameters(opt):
    """Keeps the per-step factors of `opt`'s folded update (lr, lr / bias_correction1, 1 / sqrt(bias_correction2) per group) in
    device memory (gsr_fused_adam.dynamic): the backward's launch arguments then hold nothing that changes from step to step, so a
    training step can be captured ONCE into a HIP graph and replayed, with `push_dynamic_hyperparameters` in front of each replay.
    Eager backwards keep working (they push the step's values themselves); same results bit for bit."""
    if getattr(opt, "_gsr_dynamic", None) is None:
        p0 = opt.param_groups[0]["params"][0]
        opt._gsr_dynamic = torch.zeros(_C.ADAM_DYNAMIC_FLOATS, dtype=torch.float32, device=p0.device)
    return opt._gsr_dynamic


def push_dynamic_hyperparameters(opt, advance=True):
    """Counts one optimizer step (advance=True; dense Adam's bias corrections depend on it), forms this step's factors from the
    optimizer's current learning rates and enqueues their store into the optimizer's device buffer on the current stream (the
    values travel as launch arguments: the host is free at once)."""
    groups = {g.get("name"): g for g in opt.param_groups}
    tensors = [groups[n]["params"][0] if groups[n]["params"][0].numel() else None for n in _GROUP_ORDER]
    fa, keep = _fused_adam_struct(opt, tensors, advance=advance)
    dyn = enable_dynamic_hyperparameters(opt)
    with _C.on_device(dyn.device):
        _C.check(_C.lib().gsr_adam_set_dynamic(C.byref(fa), _C.ptr(dyn), _stream()))


def prepare_for_graph_capture(device=None, slots=2):
    """Call before `torch.cuda.graph(...)` around forwards of this module (after an eager warm-up of the shapes to capture): looks
    at every status still in flight and keeps `slots` pinned status slots ready, so that the captured forward allocates no pinned
    memory and queries nothing."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    p = _ws.pool(dev)
    p.poll(wait=True)
    while len(p.status_free) < slots:
        p.status_free.append(torch.zeros(4, dtype=torch.int64).pin_memory())


def graph_status_slot(device=None):
    """(pinned status tensor, capacity, shape key) of the forward most recently captured into a HIP graph on this device:
    int64[4] = [flags, num_rendered, longest tile list, -], rewritten by every replay (third word -1 until it arrives)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    return getattr(_ws.pool(dev), "graph_status", None)


def new_tile_cull(image_height, image_width, device="cuda"):
    """Per-tile depth cut-offs of ONE view, for `GaussianRasterizer.forward(..., tile_cull=...)`: int32[tiles], all bits set = no
    limit.  Keep one per camera that is rendered again and again (the cameras of a training set): every speculative forward of the
    view updates it, every UNVERIFIED forward (mode "async") of the view emits only the instances in front of it.  A frame whose
    truncation turns out too tight flags itself (its backward is a no-op, its ticket is reported by `take_overflowed`) and the
    caller renders it again - scene_utils.Trainer does."""
    tiles = ((int(image_width) + 15) // 16) * ((int(image_height) + 15) // 16)
    return torch.full((tiles,), -1, dtype=torch.int32, device=device)


def last_ticket(device=None):
    """Ticket of the most recent speculative forward on this device (1, 2, ...): what `take_overflowed` reports."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    return _ws.pool(dev).ticket


def take_overflowed(device=None, wait=False):
    """"async" mode: tickets of the frames found truncated (binning capacity exceeded; their backward was a no-op) among the
    statuses that have arrived since the last call.  The caller runs those views again."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    p = _ws.pool(dev)
    p.poll(wait=wait)
    return p.take_overflowed()


def call_stats(device=None, wait=True):
    """Statistics of this device's rasterizer calls: `num_rendered` of the most recent frame whose count has arrived;
    frames by mode (`exact_frames`, `async_frames`, `sync_frames`); `rerendered_frames` ("exact": frames whose capacity did
    not hold and whose phase 2 was repeated), `overflow_frames` ("async": frames composited from a truncated list).
    wait=True first waits for the counts still in flight."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    p = _ws.pool(dev)
    p.poll(wait=wait)
    return dict(p.stats)


def _dump(path, rs, *tensors):
    """debug=True failure snapshot (reference README.md:168-169 `snapshot_fw.dump` / `snapshot_bw.dump`)."""
    try:
        torch.save({"settings": {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in rs._asdict().items()},
                    "tensors": [None if t is None else t.detach().cpu() for t in tensors]}, path)
        print(f"\nAn error occured in the rasterizer. Writing {path} for debugging.")
    except Exception as e:  # never mask the original error
        print(f"could not write {path}: {e}")


def _f32c(t):
    if t is None or t.numel() == 0:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _settings_struct(rs: GaussianRasterizationSettings, device):
    bg = _f32c(rs.bg.to(device))
    vm = _f32c(rs.viewmatrix.to(device))
    pm = _f32c(rs.projmatrix.to(device))
    cp = _f32c(rs.campos.to(device))
    s = _C.gsr_settings(
        int(rs.image_height), int(rs.image_width), float(rs.tanfovx), float(rs.tanfovy),
        bg.data_ptr(), float(rs.scale_modifier), vm.data_ptr(), pm.data_ptr(), int(rs.sh_degree), cp.data_ptr(),
        int(bool(rs.prefiltered)), int(bool(rs.debug)), int(bool(rs.antialiasing)))
    return s, (bg, vm, pm, cp)   # keep the tensors alive


def _gauss_struct(P, means3D, dc, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, raw_activations=False):
    def p(t):
        return None if t is None else t.data_ptr()
    return _C.gsr_gaussians(
        int(P), int(sh.shape[1]) if sh is not None else 0,
        p(means3D), p(dc), p(sh), p(colors_precomp), p(opacities), p(scales), p(rotations), p(cov3D_precomp),
        1 if raw_activations else 0)


def _stream():
    return C.c_void_p(_C.raw_stream())


class _RasterizeGaussians(torch.autograd.Function):
    """Input order = gradient order of the reference's autograd Function, with `dc` inserted
    before `sh` for the `separate_sh` call form (reference gaussian_renderer/__init__.py:90-99)."""

    @staticmethod
    def forward(ctx, means3D, means2D, dc, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                raster_settings, raw_activations=False, for_backward=True, fold=None, sh_ready_event=None, forward_mode=None,
                tile_cull=None, tile_cull_apply=True):
        """fold: a BackwardFold for THIS call's backward (kept on ctx).  sh_ready_event: a recorded torch.cuda.Event after which
        `dc` / `shs` hold this step's values (the view-sharded trainer's SH all-reduce + Adam update, in flight on another
        stream): the geometry stages run first, the stream waits for the event and only then evaluates the colours
        (gsr_forward_prepare_geometry / gsr_forward_shade).  forward_mode: "exact" | "async" | "sync" for this call (default:
        the process-wide mode, GSR_FORWARD_MODE / set_forward_mode).  tile_cull: this VIEW's per-tile depth cut-offs
        (`new_tile_cull`), updated by every speculative forward and applied by unverified ones (gsr_forward_async_culled)."""
        lib = _C.lib()
        raw_activations = bool(raw_activations) and cov3D_precomp is None
        if not means3D.is_cuda:
            raise _C.GsrError("GaussianRasterizer needs tensors on the HIP device (no CPU path)")
        dev = means3D.device
        rs = raster_settings
        P = int(means3D.shape[0])
        H, W = int(rs.image_height), int(rs.image_width)
        means3D = _f32c(means3D) if P > 0 else means3D
        dc, sh, colors_precomp = _f32c(dc), _f32c(sh), _f32c(colors_precomp)
        opacities, scales, rotations, cov3D_precomp = _f32c(opacities), _f32c(scales), _f32c(rotations), \
            _f32c(cov3D_precomp)
        # (inside forward() grad mode is always off and needs_input_grad ignores an outer torch.no_grad(): whether a backward
        # can follow is decided by the caller, rasterize_gaussians(), and arrives as `for_backward`)
        needs_grad = bool(for_backward)

        with _C.on_device(dev):
            pool = _ws.pool(dev)
            # Under HIP-graph capture (torch.cuda.graph) nothing may be queried, waited for or pinned: the frame is enqueued
            # unverified for the capacity its shape has shown (warm up eagerly first), its state lives in a Workspace of its own
            # (allocated from the graph's pool, never handed to eager calls) and its status slot is left to the graph's owner
            # (`graph_status_slot`): a replay rewrites it, and a count beyond the capacity means what it means in mode "async".
            capturing = torch.cuda.is_current_stream_capturing()
            if not capturing:
                pool.poll()                  # instance counts of earlier frames that have arrived meanwhile
                ws = pool.acquire()          # state buffers of this forward (-> backward): grow-only, recycled
            else:
                ws = _ws.Workspace(pool.device)
                ws.stream = _C.raw_stream(dev.index)
            lease = _ws.Lease(pool, ws, recycle=not capturing)
            color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
            invdepth = torch.empty(1, H, W, dtype=torch.float32, device=dev)
            radii = torch.empty(P, dtype=torch.int32, device=dev)     # every entry is written by the projection kernel
            s, keep = _settings_struct(rs, dev)
            g = _gauss_struct(P, means3D, dc, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                              raw_activations)
            geom = ws.ensure_geom(lib, P)
            img = ws.ensure_img(lib, W, H)
            ev = sh_ready_event
            split = ev is not None and colors_precomp is None
            if ev is not None and not split:
                torch.cuda.current_stream().wait_event(ev)
            evh = C.c_void_p(ev.cuda_event) if split else None
            key = (P, W, H)
            stream = _stream()
            try:
                mode = _ws.forward_mode() if forward_mode is None else forward_mode
                if mode not in _ws._MODES:
                    raise ValueError(f"forward_mode={mode!r}: expected one of {_ws._MODES}")
                if capturing:
                    if rs.debug or rs.prefiltered or key not in pool.capacity or not pool.status_free:
                        raise _C.GsrError("gsr: a forward under graph capture needs an eager warm-up of the same shape first "
                                          "(and neither debug nor prefiltered)")
                    mode = "async"
                if mode != "sync" and not rs.debug and not rs.prefiltered and P > 0:
                    # speculative: the whole frame is enqueued for the capacity this shape has shown so far.  "exact" then
                    # waits for THIS frame's count (it reaches pinned memory while the binning / compositing stages are still
                    # queued, so the device keeps working) and repeats phase 2 if the capacity did not hold; "async" does not
                    # wait: the count arrives in the pinned status slot and is looked at by a later call (pool.poll)
                    R = pool.capacity_for_frame(key)
                    binning = ws.ensure_binning(lib, P, W, H, R)
                    status = pool.status_slot()
                    status[2] = _ws.STATUS_PENDING     # (overwritten by the compositing kernel when the frame's status arrives)
                    status[3] = 0                      # (word 6: set by the compositing kernel if a truncated tile list was too short)
                    tlo = 1 if _ws.tile_local_binning(pool, key) else 0
                    count = C.c_int64(-1)
                    rerendered = False
                    verify = mode == "exact" or key not in pool.capacity     # ("async": a shape's first frame is verified too -
                    #                                                           its capacity is a guess, not an observation)
                    if tile_cull is not None:
                        tiles = ((W + 15) // 16) * ((H + 15) // 16)
                        if tile_cull.dtype != torch.int32 or tile_cull.numel() != tiles or tile_cull.device != dev \
                                or not tile_cull.is_contiguous():
                            raise _C.GsrError(f"tile_cull: expected a contiguous int32 tensor of {tiles} tiles on {dev} "
                                              "(diff_gaussian_rasterization.new_tile_cull)")
                    # lists truncated by depth only where a frame may flag itself afterwards: unverified, not under capture
                    cull_apply = tile_cull is not None and bool(tile_cull_apply) and not verify and not capturing and tlo == 1
                    _C.check(lib.gsr_forward_async_culled(C.byref(s), C.byref(g), _C.ptr(geom), geom.numel(), _C.ptr(radii),
                                                          _C.ptr(binning), binning.numel(), R, _C.ptr(img), img.numel(),
                                                          _C.ptr(color), _C.ptr(invdepth), 1 if needs_grad else 0,
                                                          1 if split else 0, evh, C.c_void_p(status.data_ptr()), tlo, stream,
                                                          C.byref(count) if verify else None, _C.ptr(tile_cull),
                                                          1 if cull_apply else 0))
                    if cull_apply:
                        pool.stats["culled_frames"] = pool.stats.get("culled_frames", 0) + 1
                    pool.ticket += 1
                    pool.stats["tile_local_frames"] = pool.stats.get("tile_local_frames", 0) + tlo
                    if verify:
                        n = int(count.value)
                        if n > R:
                            # the frame just enqueued was composited from a truncated list: same frame again, phase 2 only, on a
                            # binning state that holds it (stream-ordered behind the first attempt, which stays inside its buffers)
                            pool.stats["rerendered_frames"] += 1
                            rerendered = True
                            R = _ws._capacity_for(n)
                            binning = ws.ensure_binning(lib, P, W, H, R)
                            _C.check(lib.gsr_forward_rerender(C.byref(s), C.byref(g), _C.ptr(geom), _C.ptr(binning),
                                                              binning.numel(), R, _C.ptr(img), img.numel(), _C.ptr(color),
                                                              _C.ptr(invdepth), 1 if needs_grad else 0, tlo,
                                                              C.c_void_p(status.data_ptr()), stream))
                        pool.note(key, n)
                    pool.stats["exact_frames" if mode == "exact" else "async_frames"] += 1
                    # a verified frame's status (only its longest tile list is still of interest) is recognised by the
                    # sentinel above being overwritten: no event on the stream (a record costs ~6 us of device time between the
                    # compositing kernel and the loss); an unverified frame's is waited for in order, behind an event
                    done = None
                    if capturing:
                        pool.graph_status = (status, R, key)      # the graph's owner watches it (graph_status_slot)
                    else:
                        if not verify or rerendered:   # (a re-rendered frame writes its status twice: wait for the last one)
                            done = torch.cuda.Event()
                            done.record()
                        pool.pending.append((done, status, R, key, pool.ticket, verify))
                else:
                    # blocking read-back of num_rendered (the published rasterizer's one host synchronisation): debug mode,
                    # prefiltered=True (its "culled point" error is raised by this very call), or GSR_FORWARD_MODE=sync
                    prepare = lib.gsr_forward_prepare_geometry if split else lib.gsr_forward_prepare
                    R = _C.check(prepare(C.byref(s), C.byref(g), _C.ptr(geom), geom.numel(), _C.ptr(radii), stream))
                    pool.note(key, R)
                    pool.stats["sync_frames"] += 1
                    binning = ws.ensure_binning(lib, P, W, H, max(R, pool.capacity[key]))
                    if split:
                        # split forward: geometry stages, emission + tile sort, THEN wait for the SH update, shade, composite
                        _C.check(lib.gsr_forward_render_shade(C.byref(s), C.byref(g), _C.ptr(geom), _C.ptr(binning),
                                                              binning.numel(), R, _C.ptr(img), img.numel(), _C.ptr(color),
                                                              _C.ptr(invdepth), 1 if needs_grad else 0, evh, stream))
                    else:
                        _C.check(lib.gsr_forward_render(C.byref(s), C.byref(g), _C.ptr(geom), _C.ptr(binning),
                                                        binning.numel(), R, _C.ptr(img), img.numel(), _C.ptr(color),
                                                        _C.ptr(invdepth), 1 if needs_grad else 0, stream))
            except _C.GsrError:
                if rs.debug:   # reference README.md:168-169: with --debug a failing rasterizer call dumps its inputs
                    _dump("snapshot_fw.dump", rs, means3D, dc, sh, colors_precomp, opacities, scales, rotations,
                          cov3D_precomp)
                raise
        ctx.raster_settings = rs
        ctx.raw_activations = raw_activations
        ctx.fold = fold if needs_grad else None
        ctx.num_rendered = R                 # what the binning state was laid out for (the count itself, or the capacity)
        ctx.has = (dc is not None, sh is not None, colors_precomp is not None, scales is not None,
                   cov3D_precomp is not None)
        if needs_grad:
            ctx.lease = lease                # the state buffers stay this ctx's until it dies
        else:
            lease.release()                  # forward-only: later work on this stream may reuse them at once
        ctx.save_for_backward(means3D, dc, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, radii)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)     # an unused inverse-depth output reaches backward as None, not as zeros
        return color, radii, invdepth

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_invdepth):
        lib = _C.lib()
        (means3D, dc, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, radii) = ctx.saved_tensors
        ws = ctx.lease.ws
        if ws is None:
            raise _C.GsrError("the rasterizer's forward state is gone (backward called twice without retain_graph?)")
        geom, binning, img = ws.geom, ws.binning, ws.img
        rs = ctx.raster_settings
        R = ctx.num_rendered
        dev = means3D.device
        P = int(means3D.shape[0])
        H, W = int(rs.image_height), int(rs.image_width)
        grad_color = _f32c(grad_color) if grad_color is not None else torch.zeros(3, H, W, device=dev)
        if grad_color is None:
            grad_color = torch.zeros(3, H, W, device=dev)
        grad_invdepth = _f32c(grad_invdepth)

        def like(t, *shape):
            return torch.empty(*shape, dtype=torch.float32, device=dev) if t is not None else None

        fold = ctx.fold
        with _C.on_device(dev):
            stats = fold.stats if (fold is not None and not fold.stats_taken) else None    # (once per request)
            skip_rest = fold is not None and fold.skip_sh_rest and dc is not None and sh is not None and colors_precomp is None
            if stats is not None and (stats[0].shape[0] != P or not all(t.is_contiguous() and t.dtype == torch.float32
                                                                        for t in stats)):
                raise _C.GsrError("fold_densification_stats: statistics tensors do not match this forward's Gaussians")
            fused, split, opt = None, False, (fold.optimizer if fold is not None else None)
            if P > 0 and opt is not None and not fold.optimizer_taken and ctx.raw_activations and dc is not None \
                    and colors_precomp is None:
                split = fold.split_rows and not isinstance(opt, SparseGaussianAdam)
                # factors kept in device memory (enable_dynamic_hyperparameters): an eager backward stores this step's values
                # in front of its kernels; under graph capture nothing is counted or stored - whoever replays the graph does
                # both per replay (push_dynamic_hyperparameters)
                dynamic = getattr(opt, "_gsr_dynamic", None) is not None
                capturing = dynamic and torch.cuda.is_current_stream_capturing()
                fused = _fused_adam_struct(opt, (means3D, dc, sh, opacities, scales, rotations),
                                           rows="with_instances" if split else None, advance=not capturing)
                if fused is not None and dynamic and not capturing:
                    _C.check(lib.gsr_adam_set_dynamic(C.byref(fused[0]), _C.ptr(opt._gsr_dynamic), _stream()))
            d_means2D = torch.empty(P, 3, dtype=torch.float32, device=dev)
            if fused is None and P > 0:
                # ONE allocation for the gradients, geometry first: a data-parallel caller can exchange the four geometry tensors
                # (and dc + rest) as one contiguous span - one collective instead of four (scene_utils.parallel.GradBucket) - and
                # the dc gradient is followed by a spare row for the camera centre of the sh_rank1 exchange (dc_grad_tail_row)
                d_means3D, d_opac, d_scales, d_rot, d_dc, d_sh, d_col, d_cov = _grad_arena(dev, (
                    ((P, 3), 0), (tuple(opacities.shape) if opacities is not None else (P, 1), 0),
                    ((P, 3) if scales is not None else None, 0), ((P, 4) if rotations is not None else None, 0),
                    (tuple(dc.shape) if dc is not None else None, 3),
                    (tuple(sh.shape) if sh is not None and not skip_rest else None, 0),
                    ((P, 3) if colors_precomp is not None else None, 0), ((P, 6) if cov3D_precomp is not None else None, 0)))
            elif fused is None:
                d_means3D = torch.empty(P, 3, dtype=torch.float32, device=dev)
                d_opac = torch.empty(opacities.shape if opacities is not None else (P, 1), dtype=torch.float32, device=dev)
                d_dc = like(dc, *(dc.shape if dc is not None else ()))
                d_sh = None if skip_rest else like(sh, *(sh.shape if sh is not None else ()))
                d_col = like(colors_precomp, P, 3)
                d_scales = like(scales, P, 3)
                d_rot = like(rotations, P, 4)
                d_cov = like(cov3D_precomp, P, 6)
            else:       # the optimizer step rides in the backward: no gradient but the screen-space one is materialised
                d_means3D = d_opac = d_dc = d_sh = d_col = d_scales = d_rot = d_cov = None
            if P > 0:
                cur = _C.raw_stream()
                if ws.stream is not None and ws.stream != cur:
                    torch.cuda.current_stream().wait_stream(torch.cuda.ExternalStream(ws.stream))
                s, keep = _settings_struct(rs, dev)
                g = _gauss_struct(P, means3D, dc, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                  ctx.raw_activations)
                scratch = ws.ensure_scratch(lib, P, R)
                gr = _C.gsr_grads(*[None if t is None else t.data_ptr() for t in
                                    (d_means3D, d_means2D, d_dc, d_sh, d_col, d_opac, d_scales, d_rot, d_cov) +
                                    (stats if stats is not None else (None, None, None))])
                try:
                    if fused is not None:
                        fold.optimizer_taken = True          # (a second backward through a retained graph must not step again)
                        pool = _ws.pool(dev)
                        pool.stats["folded_backwards"] = pool.stats.get("folded_backwards", 0) + 1
                        if split:
                            # the culled rows' half: needs only the forward's state, so it runs beside the compositing backward
                            side = _side_streams.get(dev.index)
                            if side is None:
                                side = _side_streams[dev.index] = torch.cuda.Stream(device=dev)
                            side.wait_stream(torch.cuda.current_stream())
                            with torch.cuda.stream(side):
                                _C.check(lib.gsr_adam_step_culled_rows(C.byref(g), _C.ptr(geom), R, C.byref(fused[0]),
                                                                       _stream()))
                        _C.check(lib.gsr_backward_adam(C.byref(s), C.byref(g), _C.ptr(radii), _C.ptr(geom),
                                                       _C.ptr(binning), _C.ptr(img), R, _C.ptr(grad_color),
                                                       _C.ptr(grad_invdepth), _C.ptr(scratch), scratch.numel(), C.byref(gr),
                                                       C.byref(fused[0]), _stream()))
                        if split:
                            torch.cuda.current_stream().wait_stream(side)   # what follows here sees both halves of the update
                    else:
                        _C.check(lib.gsr_backward(C.byref(s), C.byref(g), _C.ptr(radii), _C.ptr(geom), _C.ptr(binning),
                                                  _C.ptr(img), R, _C.ptr(grad_color), _C.ptr(grad_invdepth),
                                                  _C.ptr(scratch), scratch.numel(), C.byref(gr), _stream()))
                except _C.GsrError:
                    if rs.debug:
                        _dump("snapshot_bw.dump", rs, means3D, dc, sh, colors_precomp, opacities, scales, rotations,
                              cov3D_precomp, grad_color, grad_invdepth, radii)
                    raise
                ws.stream = cur
            if fold is not None:
                fold.stats_taken = fold.stats_taken or stats is not None      # (P == 0: no rows, nothing to add)
                fold.sh_rest_skipped = bool(skip_rest)
        return (d_means3D, d_means2D, d_dc, d_sh, d_col, d_opac, d_scales, d_rot, d_cov, None, None, None, None, None, None, None,
                None)


def rasterize_gaussians(means3D, means2D, dc, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, raw_activations=False, fold=None, sh_ready_event=None, forward_mode=None,
                        tile_cull=None, tile_cull_apply=True):
    # forward-only render (torch.no_grad(), reference render.py:49, or no input that requires grad): the library then skips
    # what only a backward would need
    tensors = (means3D, means2D, dc, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp)
    for_backward = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)
    return _RasterizeGaussians.apply(means3D, means2D, dc, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, raw_activations, for_backward, fold, sh_ready_event,
                                     forward_mode, tile_cull, tile_cull_apply)


def pair_evaluations(raster_settings, means3D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                     cov3D_precomp=None, dc=None, raw_activations=False):
    """Pixel-Gaussian pair evaluations of one view (SURVEY.md 8(d) "FLOP model"), for the bench's pairs/s figures:
    {"num_rendered", "fwd_pairs": list entries evaluated by the compositing forward summed over pixels (instrumented build of
    the forward kernel), "fwd_blended": those of them that were blended, "bwd_pairs": sum of n_contrib (the backward replays
    entries 1..n_contrib of every pixel)}."""
    lib = _C.lib()
    rs = raster_settings
    dev = means3D.device
    P, H, W = int(means3D.shape[0]), int(rs.image_height), int(rs.image_width)
    t = [_f32c(x.detach()) if x is not None else None for x in
         (means3D, dc, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp)]
    with torch.no_grad(), torch.cuda.device(dev):
        s, keep = _settings_struct(rs, dev)
        g = _gauss_struct(P, *t, raw_activations and cov3D_precomp is None)
        geom = torch.empty(lib.gsr_geometry_state_bytes(P), dtype=torch.uint8, device=dev)
        img = torch.empty(lib.gsr_image_state_bytes(W, H), dtype=torch.uint8, device=dev)
        radii = torch.empty(P, dtype=torch.int32, device=dev)
        color = torch.empty(3, H, W, device=dev)
        invd = torch.empty(1, H, W, device=dev)
        R = _C.check(lib.gsr_forward_prepare(C.byref(s), C.byref(g), _C.ptr(geom), geom.numel(), _C.ptr(radii), _stream()))
        binning = torch.empty(lib.gsr_binning_state_bytes(P, W, H, R), dtype=torch.uint8, device=dev)
        _C.check(lib.gsr_forward_render(C.byref(s), C.byref(g), _C.ptr(geom), _C.ptr(binning), binning.numel(), R,
                                        _C.ptr(img), img.numel(), _C.ptr(color), _C.ptr(invd), 1, _stream()))
        pairs = torch.zeros(2 * H * W, dtype=torch.int32, device=dev)
        _C.check(lib.gsr_debug_count_pairs(C.byref(s), P, _C.ptr(geom), _C.ptr(binning), R, _C.ptr(pairs), _stream()))
        pT, pN = C.c_void_p(), C.c_void_p()
        lib.gsr_debug_image_views(_C.ptr(img), W, H, C.byref(pT), C.byref(pN))
        off = pN.value - img.data_ptr()
        n_contrib = img[off:off + 4 * H * W].view(torch.int32)
        out = {"num_rendered": int(R), "fwd_pairs": int(pairs[:H * W].sum(dtype=torch.int64).item()),
               "fwd_blended": int(pairs[H * W:].sum(dtype=torch.int64).item()),
               "bwd_pairs": int(n_contrib.sum(dtype=torch.int64).item())}
    return out


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """bool[P]: Gaussians in front of the near plane of this rasterizer's camera."""
        lib = _C.lib()
        with torch.no_grad():
            positions = _f32c(positions)
            P = int(positions.shape[0])
            present = torch.zeros(P, dtype=torch.uint8, device=positions.device)
            if P > 0:
                vm = _f32c(self.raster_settings.viewmatrix.to(positions.device))
                with torch.cuda.device(positions.device):
                    _C.check(lib.gsr_mark_visible(P, _C.ptr(positions), _C.ptr(vm), _C.ptr(present), _stream()))
            return present.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, dc=None, raw_activations=False, *, fold=None, sh_ready_event=None, forward_mode=None,
                tile_cull=None, tile_cull_apply=True):
        """Arguments of the reference's call (gaussian_renderer/__init__.py:90-109).  Extensions, all optional and all PER CALL
        (nothing is armed process-wide): `raw_activations=True`: `opacities`, `scales`, `rotations` are the model's RAW
        parameters; sigmoid / exp / normalize are applied inside the projection kernel and the returned gradients are w.r.t. the
        raw parameters.  `fold`: a BackwardFold (optimizer step / densification statistics / skipped dL/dsh_rest in this call's
        backward).  `sh_ready_event`: colours wait for this event.  `forward_mode`: "exact" | "async" | "sync" for this call.
        `tile_cull`: `new_tile_cull(...)` tensor of the VIEW being rendered (one per camera of a training set): tile lists
        truncated by the depth each tile saturated at when the view was last rendered (include/gsr.h gsr_forward_async_culled);
        `tile_cull_apply=False`: only keep the cut-offs up to date in this call."""
        def none_if_empty(t):
            return None if (t is None or t.numel() == 0) else t
        shs, colors_precomp, dc = none_if_empty(shs), none_if_empty(colors_precomp), none_if_empty(dc)
        scales, rotations, cov3D_precomp = none_if_empty(scales), none_if_empty(rotations), none_if_empty(cov3D_precomp)
        P = int(means3D.shape[0])
        if P > 0:
            if ((shs is None and dc is None) and colors_precomp is None) or \
                    ((shs is not None or dc is not None) and colors_precomp is not None):
                raise Exception('Please provide excatly one of either SHs or precomputed colors!')
            if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                    ((scales is not None or rotations is not None) and cov3D_precomp is not None):
                raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        return rasterize_gaussians(means3D, means2D, dc, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, self.raster_settings, raw_activations, fold, sh_ready_event, forward_mode,
                                   tile_cull, tile_cull_apply)


from .sparse_adam import SparseGaussianAdam, FusedAdam  # noqa: E402,F401   (reference train.py:37-41)
