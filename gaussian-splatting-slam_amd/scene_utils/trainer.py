"""Training-step harness with the shape of the reference loop (train.py:96-137 forward+loss+backward,
:155-168 densification statistics / densify / opacity reset, :170-179 optimizer step), used by bench.py and the tests.  It
is a CALLER of the hot path (render()); the renderer is injected so CPU tests can drive the same logic with the oracle."""
from __future__ import annotations

import torch

from .losses import training_loss_fused
from .parallel import GradBucket, reduce_densification_stats, rank1_sh_exchange


class _LazyVisibility:
    """`render()`'s "visibility_filter" (radii > 0), looked up only by the branches of a step that use it."""

    def __init__(self, pkg, value=None):
        self.pkg, self.value = pkg, value

    def get(self):
        if self.value is None:
            self.value = self.pkg["visibility_filter"]
        return self.value


class Trainer:
    def __init__(self, model, cameras, gt_images, render_fn, pipe, bg, lambda_dssim=0.2, world=1, rank=0,
                 optimizer="hip", loss="hip", depth_targets=None, depth_weight=0.0, separate_sh=False, overlap_comm=None,
                 exchange="allreduce", single_rank_group=False):
        """optimizer: "hip" (one-launch HIP Adam, default-optimizer semantics), "hip_sparse" (SparseGaussianAdam, the
        reference's accelerated choice) or "torch" (torch.optim.Adam; CPU tests).  loss: "hip" (fused SSIM kernels) or a
        callable (image, gt_image, lambda_dssim) -> scalar (the CPU tests hand in the oracle's pure-PyTorch loss).
        single_rank_group: world == 1 with an initialised process group of one rank: run the N > 1 schedule anyway (every
        collective, side stream and exchange kernel; the mean over one rank) - the rehearsal of that path on ONE GPU over RCCL."""
        self.model, self.cameras, self.gt_images = model, cameras, gt_images
        self.render_fn, self.pipe, self.bg = render_fn, pipe, bg
        self.lambda_dssim = lambda_dssim
        self.world, self.rank = world, rank
        self.distributed = world > 1 or bool(single_rank_group)
        # "hip_fused" / "hip_sparse_fused": the same two optimizers with their step folded into the rasterizer's backward
        # (gsr_backward_adam) whenever nothing sits between backward and step: one rank, one view per step, no densification
        # due.  Same arithmetic, bit for bit; the 59 floats per Gaussian of gradients never leave the kernel.
        self.fuse_step = optimizer in ("hip_fused", "hip_sparse_fused")
        # dense folded Adam: the rows WITHOUT tile instances (exact zero gradient; a third of the synthetic bench scene, most
        # of a room-scale capture) are updated on a side stream while the compositing kernels run - those are bound by VALU
        # issue and leave the HBM idle (gsr_adam_step_culled_rows; bit-identical to the unsplit update)
        self.split_rows = False          # (opt-in: measured a wash at C3 on one MI355X, see profiles/README.md)
        optimizer = {"hip_fused": "hip", "hip_sparse_fused": "hip_sparse"}.get(optimizer, optimizer)
        self.optimizer_kind = optimizer
        # like the reference, the model owns the optimizer and the densification statistics (gaussian_model.py:155-176)
        self.optimizer = model.training_setup(optimizer=optimizer)
        if not callable(loss) and loss != "hip":
            raise ValueError("loss: 'hip' or a callable (the pure-PyTorch loss is test infrastructure: oracle/loss_oracle.py)")
        self.loss_fn = loss if callable(loss) else training_loss_fused
        self._loss_arg = loss
        # (an initialised process group alone changes nothing: only `single_rank_group` makes a one-rank trainer run collectives)
        self._force_collectives = bool(single_rank_group) and world <= 1
        self.bucket = GradBucket(self._arena_order_params()) if self.distributed else None
        # N > 1 gradient exchange: "allreduce" (one all-reduce per leaf, every rank runs the whole update),
        # "visible_rows" (the same, restricted to the rows some rank saw), "sharded" (reduce-scatter -> Adam on a 1/N row shard
        # -> all-gather of the parameters; dense optimizers, no densification yet: the moments live per shard)
        # "sh_rank1" (one view per rank per step): the 11 geometry floats are all-reduced, the 48 SH floats are rebuilt from an
        # all-gather of dL/df_dc + camera centres (parallel.rank1_sh_exchange): 161 instead of 413 B per Gaussian at N = 8
        if exchange not in ("allreduce", "visible_rows", "sharded", "sh_rank1"):
            raise ValueError(exchange)
        self.exchange = exchange if self.distributed else "allreduce"
        self.sharded = None
        if self.exchange == "sharded":
            if optimizer not in ("hip", "torch"):
                raise ValueError("exchange='sharded' needs a dense optimizer ('hip' or 'torch')")
            from .parallel import ShardedStep
            if optimizer == "torch":
                mk = lambda groups: torch.optim.Adam(groups, lr=0.0, eps=1e-15)      # noqa: E731
            else:
                from diff_gaussian_rasterization import FusedAdam
                mk = lambda groups: FusedAdam(groups, lr=0.0, eps=1e-15)             # noqa: E731
            self._mk_sharded = mk
            self.sharded = ShardedStep(model, mk, world, rank)
            self.optimizer = self.sharded.optimizer
        self.depth_targets, self.depth_weight = depth_targets, depth_weight
        # reference train.py:106 passes separate_sh=SPARSE_ADAM_AVAILABLE: dc / rest go to the rasterizer unconcatenated
        self.separate_sh = separate_sh
        # Overlap of the cross-rank exchange with the next step's geometry stages: needs the HIP device, the
        # HIP optimizers (they step parameter groups separately) and dc / rest passed unconcatenated (separate_sh), because a
        # torch.cat of the SH tensors would read them on the main stream while their update is still in flight.
        can_overlap = optimizer in ("hip", "hip_sparse") and separate_sh and model.get_xyz.is_cuda and \
            self.exchange in ("allreduce", "sh_rank1")
        # default: on for N > 1.  On one GPU it can be requested, but it buys nothing (measured 2.22 vs 2.21 ms/step at C3: the
        # Adam kernel fills the machine, the small geometry kernels just queue behind it); what it hides is COMMUNICATION.
        self.overlap_comm = (can_overlap and self.distributed) if overlap_comm is None else (bool(overlap_comm) and can_overlap)
        self.side_stream = torch.cuda.Stream(device=model.get_xyz.device) if self.overlap_comm else None
        if self.overlap_comm:
            self.optimizer.init_state()      # moments live in the main stream's pool, never the side stream's
        self.densify = None          # schedule dict once enable_densification() is called
        self.iteration = 0
        self._one = None                 # cached dL/dloss = 1 for loss.backward()
        self.last = {}
        self._rank1_fused = False
        self.rank1_fuse_adam = True      # exchange "sh_rank1": let the rebuilding kernel apply the SH groups' Adam step (dense HIP Adam)
        self._ticket_view = {}           # forward mode "async": rasterizer ticket -> view of the frames still unverified
        self.tile_cull = None            # enable_tile_cull(): view -> per-tile depth cut-offs (lists truncated by depth)
        self.graph_replay = False        # enable_graph_replay(): the single-view step as ONE HIP graph launch
        self._graph = None
        self.rerun_views = 0             # ... and how many truncated frames were run again

    def _arena_order_params(self):
        """The six leaves in the order the rasterizer's backward lays their gradients out (geometry first): GradBucket's span over
        them is then a view of that allocation, no copy."""
        m = self.model
        return [m._xyz, m._opacity, m._scaling, m._rotation, m._features_dc, m._features_rest]

    # statistics live in the model (reference: GaussianModel.xyz_gradient_accum / denom / max_radii2D)
    @property
    def xyz_gradient_accum(self):
        return self.model.xyz_gradient_accum

    @property
    def denom(self):
        return self.model.denom

    @property
    def max_radii2D(self):
        return self.model.max_radii2D

    def enable_tile_cull(self, on=True):
        """Tile lists truncated by depth (include/gsr.h gsr_forward_async_culled): every view of the training set keeps the depth
        each of its tiles saturated at when it was last rendered (+ a margin); the next UNVERIFIED render of the view (forward mode
        "async": folded optimizer, one view per step, one rank) emits only the instances in front of it.  Where the scene
        saturates early this removes most of the step's R-proportional work.  A frame whose truncation was too tight flags
        itself - its backward, optimizer step included, is a no-op on the device - and is run again untruncated before the next
        view is touched (the machinery of forward mode "async"): parameters, moments and statistics equal the untruncated
        run's bit for bit (tests/test_tile_cull_gpu.py).  Verified steps only keep the cut-offs up to date."""
        self.tile_cull = {} if on else None
        # back-off per view: a frame that flagged itself costs a whole extra step, so the view then renders untruncated for 2, 4,
        # ... 16 visits (its cut-offs keep being refreshed) before truncation is tried again; one clean truncated visit resets it.
        # Where the scene changes faster than the views come round (a growing SLAM-style map) this keeps the cost at a few re-runs.
        self._cull_wait, self._cull_penalty = {}, {}
        # ... and a governor over all views: more than 2 flagged among 50 truncated frames (a scene that still moves fast: training
        # from scratch with densification, config 5) switches truncation off for the next 1000 iterations, 2000 the next time, ...
        # (a clean window halves the term again)
        self._cull_window, self._cull_off_until, self._cull_off_len = [0, 0], -1, 1000
        self.cull_adaptive = True        # (False: truncate every eligible frame - the sweeps want every frame tried)

    def enable_densification(self, extent, from_iter=500, until_iter=15000, interval=100, opacity_reset_interval=3000,
                             grad_threshold=0.0002, min_opacity=0.005, seed=0, max_gaussians=None):
        """Schedule and thresholds of reference train.py:155-168 / arguments/__init__.py:84-90.  `max_gaussians` (not in the
        reference; for the synthetic growth benchmark): no further densification once the model has that many rows."""
        self.densify = dict(extent=extent, from_iter=from_iter, until_iter=until_iter, interval=interval,
                            reset=opacity_reset_interval, thr=grad_threshold, min_opacity=min_opacity, seed=seed,
                            max_gaussians=max_gaussians)

    def step(self, view_idx, forward_mode=None):
        """One optimizer step.  `view_idx`: one view (the reference's batch-1 step) or a list of views whose gradients are
        accumulated locally before the single cross-rank exchange and the single Adam step (gradient accumulation).
        `forward_mode`: the rasterizer's forward mode for this step's renders (default: the process-wide one)."""
        views = list(view_idx) if isinstance(view_idx, (list, tuple)) else [view_idx]
        if self.graph_replay and forward_mode is None and len(views) == 1 and self._graph_step(views[0]):
            return self.last
        return self._eager_step(views, forward_mode)

    def _render_kwargs(self, n_views, mode, unverified):
        """Per-call extensions handed to render() on the HIP device (all state of a step's hand-offs lives in these objects and on
        the autograd ctx of the call they are given to - nothing is armed process-wide): -> (kwargs, BackwardFold | None, whether
        the optimizer step was asked to ride in the backward)."""
        if not self.model.get_xyz.is_cuda:
            return {}, None, False        # (the CPU tests inject the oracle as renderer: the reference's plain call form)
        import diff_gaussian_rasterization as dgr
        m = self.model
        due = self._densify_due(self.iteration + 1)
        want_fold = self.fuse_step and not self.distributed and n_views == 1 and not due and self.separate_sh
        # exchange "sh_rank1" with the dense HIP Adam: this rank's dL/df_rest is not exchanged (rebuilt from the ranks' dL/df_dc)
        # and, unless a densification sits between backward and step, the two SH groups' Adam step rides in the rebuilding
        # kernel: the backward need not write those 180 B per Gaussian at all
        self._rank1_fused = (self.rank1_fuse_adam and self.exchange == "sh_rank1" and self.distributed and n_views == 1 and
                             self.optimizer_kind == "hip" and self.separate_sh and not due)
        # this view's densification statistics ride in the rasterizer's backward (one pass and one launch less)
        fold = dgr.BackwardFold(optimizer=self.optimizer if want_fold else None, split_rows=self.split_rows,
                                stats=(m.xyz_gradient_accum, m.denom, m.max_radii2D), skip_sh_rest=self._rank1_fused)
        kw = {"fold": fold}
        if mode == "async" and not unverified:
            kw["forward_mode"] = "exact"          # this step's frames are verified although the process default is not
        elif mode is not None and mode != dgr.forward_mode():
            kw["forward_mode"] = mode
        ev = getattr(self, "_pending_sh_event", None)
        if ev is not None:                        # the previous step's SH update is still in flight on the side stream
            kw["sh_ready_event"] = ev
        return kw, fold, want_fold

    def _tile_cull_of(self, v):
        """This view's cut-off tensor (created on first use: no limit anywhere)."""
        import diff_gaussian_rasterization as dgr
        t = self.tile_cull.get(v)
        if t is None:
            cam = self.cameras[v]
            t = self.tile_cull[v] = dgr.new_tile_cull(cam.image_height, cam.image_width, self.model.get_xyz.device)
        return t

    def _eager_step(self, views, forward_mode=None):
        is_hip = self.model.get_xyz.is_cuda
        mode = None
        unverified = False
        if is_hip:
            import diff_gaussian_rasterization as dgr
            mode = forward_mode if forward_mode is not None else dgr.forward_mode()
        if mode == "async":
            # the status of the previous step's frame is waited for HERE (it left the device when that frame's compositing kernel
            # started, i.e. long ago unless the host runs more than a step ahead): a truncated frame is run again before the
            # next view is touched, so the updates keep the order of the default mode
            self._rerun_truncated_frames(wait=True)
            # A truncated frame is only harmless where its backward IS the whole step: optimizer (and statistics) folded into it,
            # so the device-side guard leaves parameters, moments and statistics untouched and the view can simply be run again.
            # Any other step - optimizer as its own launch (a zero-gradient Adam step still moves parameters), a densification
            # due, several views, N > 1 - verifies its frames like the default mode does.
            unverified = (self.fuse_step and not self.distributed and len(views) == 1 and self.separate_sh
                          and not self._densify_due(self.iteration + 1) and not getattr(self, "_fold_refused", False))
        self._rank1_fused = False
        for n, v in enumerate(views):
            cam = self.cameras[v]
            kw, fold, want_fold = self._render_kwargs(len(views), mode, unverified)
            if self.tile_cull is not None and is_hip:
                kw["tile_cull"] = self._tile_cull_of(v)      # (applied by unverified forwards, kept up to date by all)
                if not self.cull_adaptive:
                    pass
                elif self.iteration < self._cull_off_until:
                    kw["tile_cull_apply"] = False
                elif self._cull_wait.get(v, 0) > 0:
                    self._cull_wait[v] -= 1
                    kw["tile_cull_apply"] = False
                elif unverified:
                    self._cull_penalty[v] = max(0, self._cull_penalty.get(v, 0) - 1)   # (decays with every truncated visit)
                    self._cull_window[0] += 1
                    if self._cull_window[0] >= 50 or self._cull_window[1] > 2:
                        if self._cull_window[1] > 2:
                            self._cull_off_until = self.iteration + self._cull_off_len
                            self._cull_off_len = min(2 * self._cull_off_len, 64000)
                        else:
                            self._cull_off_len = max(1000, self._cull_off_len // 2)
                        self._cull_window = [0, 0]
            pkg = self.render_fn(cam, self.model, self.pipe, self.bg, separate_sh=self.separate_sh, **kw)
            if "sh_ready_event" in kw:
                self._pending_sh_event = None     # (that forward made its stream wait for it)
            if unverified:
                self._ticket_view[dgr.last_ticket(self.model.get_xyz.device)] = v
            image, vsp, radii = pkg["render"], pkg["viewspace_points"], pkg["radii"]
            vis = _LazyVisibility(pkg)       # `radii > 0`: only materialised by the branches that use it
            loss = self.loss_fn(image, self.gt_images[v], self.lambda_dssim)
            if self.depth_weight > 0 and self.depth_targets is not None:
                if pkg["depth"].is_cuda and not callable(self._loss_arg):
                    from fused_ssim import l1_mean_loss
                    loss = loss + l1_mean_loss(pkg["depth"], self.depth_targets[v], self.depth_weight)
                else:
                    loss = loss + self.depth_weight * torch.abs(pkg["depth"] - self.depth_targets[v]).mean()
            if len(views) > 1:
                loss = loss / len(views)
            if self._one is None or self._one.device != loss.device:
                self._one = torch.ones((), dtype=loss.dtype, device=loss.device)
            loss.backward(gradient=self._one)          # .grad accumulates; (a cached seed: no fill launch per step)
            # what the backward took of the request (a renderer that does not hand `fold` to the rasterizer - a model wrapper
            # without raw parameters, the python-SH branch ... - leaves everything False: plain calls below)
            folded = fold is not None and fold.optimizer_taken
            self._rank1_fused = self._rank1_fused and fold is not None and fold.sh_rest_skipped
            if want_fold and not folded and unverified:
                # this frame ran unverified and its update is about to happen as a launch of its own - look at its status NOW,
                # and from here on verify every frame
                self._fold_refused = True
                t = dgr.last_ticket(self.model.get_xyz.device)
                if t in dgr.take_overflowed(self.model.get_xyz.device, wait=True):
                    self._ticket_view.pop(t, None)
                    self.optimizer.zero_grad(set_to_none=True)
                    return self._eager_step(views, forward_mode)    # (truncated: zero gradients, nothing applied yet - once more, verified)
            if fold is None or not fold.stats_taken:
                with torch.no_grad():
                    # densification statistics are per-view: norm BEFORE any cross-rank reduction
                    self.model.add_densification_stats(vsp, vis.get(), radii)                # train.py:159-160
            if n + 1 < len(views):
                vis_any = vis.get() if n == 0 else (vis_any | vis.get())
        if len(views) > 1:
            vis = _LazyVisibility(None, vis_any | vis.get())
        self.iteration += 1
        self.last = dict(loss=loss.detach(), image=image.detach(), radii=radii)
        if len(views) == 1 and folded and self.densify is None:
            return self.last            # parameters already updated by the backward; no gradient was stored, nothing to zero
        if self.distributed and self.optimizer_kind == "hip_sparse":
            # SparseGaussianAdam updates the rows visible in "the" view; with views sharded over ranks that is the UNION of
            # the ranks' visibility masks (a Gaussian seen by any rank has a non-zero averaged gradient) - otherwise the
            # replicas would drift apart
            import torch.distributed as dist
            v8 = vis.get().to(torch.uint8)
            dist.all_reduce(v8, op=dist.ReduceOp.MAX)
            vis = _LazyVisibility(None, v8.bool())
        # the rank-one form of the SH exchange holds for ONE view's gradient (one direction per Gaussian)
        rank1_cam = self.cameras[views[0]].camera_center if (self.exchange == "sh_rank1" and len(views) == 1) else None
        if self.overlap_comm and not self._densify_due():
            self._exchange_and_step_overlapped(vis.get() if self.optimizer_kind == "hip_sparse" else None, radii, rank1_cam)
            return self.last
        if self.sharded is not None:
            # The same iteration the all-reduce schedule runs (reference order: train.py:155-168 densify / reset BETWEEN backward
            # and optimizer.step, the replaced Parameters carry no gradient and are skipped by that step): a densification replaces
            # every parameter - nothing is exchanged or applied; an opacity reset replaces `_opacity` alone - the other five step.
            grow, reset = self._densify_plan()
            if grow:
                for p in self.model.parameters():
                    p.grad = None
            else:
                self.sharded.step(skip=(self.model._opacity,) if reset else ())
            if grow or reset:
                with torch.no_grad():
                    self._maybe_densify(radii)
            return self.last
        with torch.no_grad():
            if self.bucket is not None and rank1_cam is not None:
                m = self.model
                self.bucket.all_reduce_mean(self.world, params=[m._xyz, m._opacity, m._scaling, m._rotation],
                                            force=self._force_collectives)
                rank1_sh_exchange(m._xyz, m._features_dc, m._features_rest, rank1_cam, m.active_sh_degree, self.world,
                                  optimizer=self.optimizer if self._rank1_fused else None)
            elif self.bucket is not None:
                self.bucket.all_reduce_mean(self.world, visible=vis.get() if self.exchange == "visible_rows" else None,
                                            force=self._force_collectives)
            if self.densify is not None:
                # the reference densifies between backward and the optimizer step (train.py:155-168 before :170): the
                # replaced Parameters carry no gradient, so the step that follows skips them, exactly like there
                self._maybe_densify(radii)
            if folded:
                pass                     # (densification schedule active but not due: the backward already applied the step)
            elif self.optimizer_kind == "hip_sparse":
                self.optimizer.step(vis.get(), radii.shape[0])                                 # train.py:173-175
            else:
                self.optimizer.step()
            self.optimizer.zero_grad(set_to_none=True)
        return self.last

    @staticmethod
    def _unverified_mode():
        """The process-wide forward mode is "async" (steps may run unverified frames; see _eager_step)."""
        import diff_gaussian_rasterization as dgr
        return dgr.forward_mode() == "async"

    def _rerun_truncated_frames(self, wait=False):
        """Forward mode "async" only.  A frame that had more tile instances than its binning state held was composited from a
        truncated list; the rasterizer's backward then did nothing (zero gradients; with the optimizer folded in: parameters,
        moments and statistics bit-unchanged) and reports the frame's ticket once its status has arrived.  Such a view is run
        again here, verified, as one more iteration - before the next view is touched."""
        import diff_gaussian_rasterization as dgr
        dev = self.model.get_xyz.device
        for t in dgr.take_overflowed(dev, wait=wait):
            v = self._ticket_view.pop(t, None)
            if v is None:
                continue
            if self.fuse_step and not self.distributed:
                # the folded step that did not happen was counted on the host: the bias correction must not see it
                for st in self.optimizer.state.values():
                    if "step" in st and float(st["step"]) > 0:
                        st["step"] -= 1
            self.iteration -= 1              # ... and neither may the iteration count (densification / reset schedule)
            self.rerun_views += 1
            if self.tile_cull is not None:   # (a truncation that was too tight, most likely: back off on this view)
                p = self._cull_penalty[v] = min(16, max(2, 2 * self._cull_penalty.get(v, 0)))
                self._cull_wait[v] = p
                self._cull_window[1] += 1
            self._eager_step([v], forward_mode="exact")
        if len(self._ticket_view) > 256:      # statuses arrive in order: anything older than the newest 64 is long verified
            for t in sorted(self._ticket_view)[:-64]:
                del self._ticket_view[t]

    def _densify_due(self, it=None):
        d, it = self.densify, (self.iteration if it is None else it)
        if d is None or it >= d["until_iter"]:
            return False
        grow = it > d["from_iter"] and it % d["interval"] == 0 and not self._at_capacity()
        return grow or it % d["reset"] == 0

    def _densify_plan(self, it=None):
        """(densify_and_prune runs, reset_opacity runs) at iteration `it` (default: the current one) - what _maybe_densify does."""
        d, it = self.densify, (self.iteration if it is None else it)
        if d is None or it >= d["until_iter"]:
            return False, False
        return (it > d["from_iter"] and it % d["interval"] == 0 and not self._at_capacity()), it % d["reset"] == 0

    def _at_capacity(self):
        m = self.densify.get("max_gaussians") if self.densify else None
        return m is not None and int(self.model.get_xyz.shape[0]) >= m

    @torch.no_grad()
    def _exchange_and_step_overlapped(self, vis, radii, rank1_cam=None):
        """Synchronous data parallelism, same result as the plain path, different schedule: the SH gradients (f_dc, f_rest:
        48 of the 59 floats per Gaussian) are all-reduced and applied on a side stream; the main stream exchanges and applies
        the geometry gradients (11 floats) and goes straight on to the next step, whose rasterizer runs projection, depth
        sort, emission and tile sort before it waits for the SH update (diff_gaussian_rasterization.defer_sh_until)."""
        m = self.model
        geo, geo_names = [m._xyz, m._opacity, m._scaling, m._rotation], ("xyz", "opacity", "scaling", "rotation")
        sh, sh_names = [m._features_dc, m._features_rest], ("f_dc", "f_rest")
        main, side = torch.cuda.current_stream(), self.side_stream
        # Collectives of one communicator run in ISSUE order (RCCL keeps one internal stream per communicator): the small
        # geometry exchange goes first so that the main stream only ever waits for it; the SH exchange queues behind it.
        if self.bucket is not None:
            self.bucket.all_reduce_mean(self.world, params=geo, force=self._force_collectives)
        side.wait_stream(main)                                   # gradients complete (and the geometry exchange issued)
        with torch.cuda.stream(side):
            for p in sh:
                if p.grad is not None:
                    p.grad.record_stream(side)
            if vis is not None:
                vis.record_stream(side)
            if self.bucket is not None and rank1_cam is not None:
                # the SH gradients are rebuilt from the ranks' dL/df_dc and the positions the forwards SAW: the main stream's
                # geometry update must not overtake that read, so the side stream takes a copy first (12 B per Gaussian)
                xyz_seen = m._xyz.detach().clone()
                snap = torch.cuda.Event()
                snap.record(side)
                main.wait_event(snap)
                stepped = rank1_sh_exchange(xyz_seen, m._features_dc, m._features_rest, rank1_cam, m.active_sh_degree,
                                            self.world, optimizer=self.optimizer if self._rank1_fused else None)
                if stepped:
                    sh_names = ()                     # the rebuilding kernel applied the SH groups' step itself
            elif self.bucket is not None:
                self.bucket.all_reduce_mean(self.world, params=sh, force=self._force_collectives)
            if not sh_names:
                pass
            elif self.optimizer_kind == "hip_sparse":
                self.optimizer.step(vis, radii.shape[0], only=sh_names)
            else:
                self.optimizer.step(only=sh_names)
            ev = torch.cuda.Event()
            ev.record(side)
        if self.optimizer_kind == "hip_sparse":
            self.optimizer.step(vis, radii.shape[0], only=geo_names)
        else:
            self.optimizer.step(only=geo_names)
        # handed to this trainer's NEXT render as `sh_ready_event` (per call: a render of somebody else in between - a viewer,
        # an evaluation - is not touched by it and must call finish() before it reads the SH parameters)
        self._pending_sh_event = ev
        self.optimizer.zero_grad(set_to_none=True)

    # ---- the single-view step as one HIP graph launch (launch-bound scenes: a few 10 k Gaussians, small images) ----
    def enable_graph_replay(self, on=True, warmup=2):
        """One rank, one view per step, optimizer folded into the backward ("hip_fused" / "hip_sparse_fused"), HIP loss: after
        `warmup` eager steps per (model size, SH degree, image size) the whole step - forward, loss, backward, Adam, densification
        statistics - is captured ONCE (torch.cuda.graph over the library's launches) and replayed with one host call per step.
        What changes between steps lives in device memory: camera matrices and ground truth are copied into static buffers, the
        optimizer's per-step factors are pushed in front of each replay (push_dynamic_hyperparameters), the forward runs
        unverified for the capacity the shape has shown.  The status of step k is looked at before step k + 1 is launched: a
        frame beyond the capacity was a no-op on the device (as in forward mode "async"); it is then run again eagerly, the
        capacity is raised and the graph captured anew.  Steps the graph does not cover (several views, a densification due,
        N > 1) run eagerly.  Same parameters, moments and statistics as the eager loop, bit for bit."""
        if on and not (self.fuse_step and not self.distributed and self.separate_sh and self.model.get_xyz.is_cuda
                       and self._gt_is_hip_loss()):
            raise ValueError("graph replay needs one rank, a *_fused optimizer, separate_sh and the HIP loss on the HIP device")
        self.graph_replay, self._graph_warmup = bool(on), int(warmup)
        if on:
            self.optimizer.init_state()      # the moments' addresses are part of a captured step's signature: create them now
        self._graph, self._graph_sig, self._graph_warm = None, None, {}
        self.graph_stats = dict(captures=0, replays=0, eager_steps=0, overflow_reruns=0)

    def _gt_is_hip_loss(self):
        return self.loss_fn is training_loss_fused

    def _graph_signature(self, cam):
        """Everything a captured step has baked into its launch arguments: shapes, the camera's intrinsics, the scalars of the
        loss, and the ADDRESS of every tensor the graph reads or writes in place - all six parameters, both Adam moments of
        each, the three statistics tensors.  (reset_opacity replaces `_opacity` and its moments and nothing else; a densification
        replaces everything: either way the signature changes and the stale graph is never replayed - ADVICE r3.)"""
        m = self.model
        ptrs = []
        for p in m.parameters():
            st = self.optimizer.state.get(p, {})
            ptrs += [p.data_ptr(), st["exp_avg"].data_ptr() if "exp_avg" in st else 0,
                     st["exp_avg_sq"].data_ptr() if "exp_avg_sq" in st else 0]
        ptrs += [m.xyz_gradient_accum.data_ptr(), m.denom.data_ptr(), m.max_radii2D.data_ptr()]
        has_depth = self.depth_weight > 0 and self.depth_targets is not None
        return (int(m.get_xyz.shape[0]), int(m.active_sh_degree), int(cam.image_height), int(cam.image_width),
                float(cam.FoVx), float(cam.FoVy), float(self.lambda_dssim), float(self.depth_weight) if has_depth else 0.0,
                self.bg.data_ptr(),
                *ptrs)

    def _drop_graph(self):
        """The model's tensors were replaced (densification, pruning, opacity reset): the captured step points at the old ones."""
        if getattr(self, "_graph", None) is not None:
            self._graph_settle()
        self._graph = self._graph_sig = None

    def _graph_step(self, v):
        """-> True if the step was done by a replay (or its eager stand-in), False: let the eager path do it."""
        import diff_gaussian_rasterization as dgr
        if self._densify_due(self.iteration + 1) or self.split_rows or int(self.model.get_xyz.shape[0]) == 0:
            self._graph_settle()          # (an empty model - everything pruned - has nothing to capture: eager)
            return False
        cam = self.cameras[v]
        sig = self._graph_signature(cam)
        if self._graph is None or self._graph_sig != sig:
            self._graph_settle()
            if self._graph_warm.get(sig, 0) < self._graph_warmup:       # the shape's capacity has to be known first
                self._graph_warm[sig] = self._graph_warm.get(sig, 0) + 1
                self.graph_stats["eager_steps"] += 1
                return False
            self._graph_capture(sig, cam, v)
        if not self._graph_settle():          # the step before this one had to be run again eagerly: graph gone, start over
            return self._graph_step(v)
        g = self._graph_state
        for dst, src in zip(g["cam_tensors"], (cam.world_view_transform, cam.full_proj_transform, cam.camera_center)):
            dst.copy_(src, non_blocking=True)
        g["gt"].copy_(self.gt_images[v], non_blocking=True)
        if g["depth"] is not None:
            g["depth"].copy_(self.depth_targets[v], non_blocking=True)
        g["status"][2] = -1                                   # (rewritten by the replay's compositing kernel)
        dgr.push_dynamic_hyperparameters(self.optimizer, advance=True)
        self._graph.replay()
        self.iteration += 1
        self._graph_inflight = v
        self.graph_stats["replays"] += 1
        self.last = dict(loss=g["loss"], image=g["image"], radii=g["radii"])
        return True

    def _graph_capture(self, sig, cam, v):
        import copy
        import diff_gaussian_rasterization as dgr
        m, dev = self.model, self.model.get_xyz.device
        cam_s = copy.copy(cam)
        cam_s.world_view_transform = cam.world_view_transform.clone()
        cam_s.full_proj_transform = cam.full_proj_transform.clone()
        cam_s.camera_center = cam.camera_center.clone()
        gt = self.gt_images[v].clone()
        depth = self.depth_targets[v].clone() if (self.depth_weight > 0 and self.depth_targets is not None) else None
        if self._one is None or self._one.device != dev:
            self._one = torch.ones((), dtype=torch.float32, device=dev)
        dgr.enable_dynamic_hyperparameters(self.optimizer)
        dgr.prepare_for_graph_capture(dev)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        fold = dgr.BackwardFold(optimizer=self.optimizer, stats=(m.xyz_gradient_accum, m.denom, m.max_radii2D))
        with torch.cuda.graph(graph):
            pkg = self.render_fn(cam_s, m, self.pipe, self.bg, separate_sh=True, fold=fold)
            loss = self.loss_fn(pkg["render"], gt, self.lambda_dssim)
            if depth is not None:
                from fused_ssim import l1_mean_loss
                loss = loss + l1_mean_loss(pkg["depth"], depth, self.depth_weight)
            loss.backward(gradient=self._one)
        if not (fold.optimizer_taken and fold.stats_taken):
            raise RuntimeError("graph capture: the rasterizer's backward did not take the optimizer / statistics hand-off")
        status, cap, key = dgr.graph_status_slot(dev)
        self._graph, self._graph_sig = graph, sig
        self._graph_state = dict(cam_tensors=(cam_s.world_view_transform, cam_s.full_proj_transform, cam_s.camera_center), gt=gt,
                                 depth=depth, loss=loss.detach(), image=pkg["render"].detach(), radii=pkg["radii"], status=status,
                                 capacity=cap, key=key, cam=cam_s)
        self._graph_inflight = None
        self.graph_stats["captures"] += 1

    def _graph_settle(self):
        """Waits for the status of the replay in flight (if any).  -> True: fine.  False: that frame had more tile instances than
        the graph's binning state holds - its backward was a no-op on the device - so the step was counted back, run again
        eagerly (verified), and the graph dropped (the next step captures a new one for the raised capacity)."""
        import diff_gaussian_rasterization as dgr
        from diff_gaussian_rasterization import _workspace as ws
        v = getattr(self, "_graph_inflight", None)
        if self._graph is None or v is None:
            return True
        g = self._graph_state
        ws._wait_bounded(ws._StatusArrived(g["status"]), "the status of the replayed frame")
        self._graph_inflight = None
        flags, R = int(g["status"][0]), int(g["status"][1])
        if flags & ws.STATUS_SORT_TIMEOUT:
            raise dgr._C.GsrError("gsr: a radix-sort look-back wait timed out in the replayed frame (tile lists not to be trusted)")
        pool = ws.pool(self.model.get_xyz.device)
        pool.stats["num_rendered"] = R
        if R <= g["capacity"] and R >= 0:
            return True
        # truncated: nothing was updated (gsr_overflowed); undo the host-side count, learn the capacity, redo the view eagerly
        for st in self.optimizer.state.values():
            if "step" in st and float(st["step"]) > 0 and self.optimizer_kind == "hip":
                st["step"] -= 1
        self.iteration -= 1
        pool.note(g["key"], R)
        self._graph, self._graph_sig = None, None
        self.graph_stats["overflow_reruns"] += 1
        self._eager_step([v], forward_mode="exact")
        return False

    def finish(self):
        """Make the main stream wait for an SH update still in flight (call before reading the parameters outside step())."""
        ev = getattr(self, "_pending_sh_event", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
        if self._graph is not None:
            self._graph_settle()
        if self.model.get_xyz.is_cuda and self._unverified_mode():
            self._rerun_truncated_frames(wait=True)      # nothing truncated is left behind

    def _maybe_densify(self, radii):
        d, it = self.densify, self.iteration
        if it >= d["until_iter"]:
            return
        changed = False
        grow, reset = self._densify_plan(it)
        if self.sharded is not None and (grow or reset):
            # exchange "sharded": the Adam moments exist per row shard.  The row surgery wants whole tensors: gather them (472 MB at
            # 1 M Gaussians, once per densification interval), hand them to the model's own - in this mode otherwise stateless -
            # optimizer, whose state densify_and_prune / reset_opacity carry through, and slice the result again below
            for (p, _, _), m in zip(self.sharded.items, self.sharded.full_moments()):
                if m is not None:
                    self.model.optimizer.state[p] = {"step": m[2].clone(), "exp_avg": m[0], "exp_avg_sq": m[1]}
        if it > d["from_iter"] and it % d["interval"] == 0 and not self._at_capacity():
            # per-view statistics -> identical on all ranks, so every rank takes the same decisions
            reduce_densification_stats(self.model.xyz_gradient_accum, self.model.denom, self.model.max_radii2D, self.world,
                                       force=self._force_collectives)
            size_threshold = 20 if it > d["reset"] else None
            self.last["densify"] = self.model.densify_and_prune(d["thr"], d["min_opacity"], d["extent"], size_threshold,
                                                                 radii, seed=d["seed"] + it)
            changed = True
        if it % d["reset"] == 0:
            self.model.reset_opacity()
            changed = True
        if changed and self.sharded is not None:
            from .parallel import ShardedStep
            self.sharded = ShardedStep(self.model, self._mk_sharded, self.world, self.rank, init_from=self.model.optimizer)
            self.optimizer = self.sharded.optimizer
            self.model.optimizer.state.clear()      # (the full-size moments were only passing through)
        if changed and self.bucket is not None:     # the replaced Parameters are new objects (and carry no gradient yet)
            self.bucket = GradBucket(self._arena_order_params())
        if changed and self.graph_replay:
            self._drop_graph()                      # (explicitly: not left to the signature alone)
        if changed and self.tile_cull:
            # a densification / pruning / opacity reset moves every tile's saturation depth at once: cut-offs learnt before it would
            # flag a third of the frames that follow (measured on config 5); every view relearns its own on its next render
            for t in self.tile_cull.values():
                t.fill_(-1)
