"""Training-step harness with the shape of the reference loop (train.py:96-137 forward+loss+backward,
:159-160 densification statistics, :170-179 optimizer step), used by bench.py and the tests.  It is a CALLER of the
hot path (render()); the renderer is injected so CPU tests can drive the same logic with the oracle."""
from __future__ import annotations

import torch

from .losses import training_loss, training_loss_fused
from .parallel import GradBucket


class Trainer:
    def __init__(self, model, cameras, gt_images, render_fn, pipe, bg, lambda_dssim=0.2, world=1, rank=0,
                 optimizer="hip", loss="hip", depth_targets=None, depth_weight=0.0, separate_sh=False):
        """optimizer: "hip" (one-launch HIP Adam, default-optimizer semantics), "hip_sparse" (SparseGaussianAdam, the
        reference's accelerated choice) or "torch" (torch.optim.Adam; CPU tests).  loss: "hip" (fused SSIM kernels) or
        "torch" (pure-PyTorch ssim; CPU tests)."""
        self.model, self.cameras, self.gt_images = model, cameras, gt_images
        self.render_fn, self.pipe, self.bg = render_fn, pipe, bg
        self.lambda_dssim = lambda_dssim
        self.world, self.rank = world, rank
        dev = model.get_xyz.device
        self.optimizer_kind = optimizer
        if optimizer == "torch":
            self.optimizer = torch.optim.Adam(model.param_groups(), lr=0.0, eps=1e-15)   # scene/gaussian_model.py:170
        elif optimizer == "hip":
            from diff_gaussian_rasterization import FusedAdam
            self.optimizer = FusedAdam(model.param_groups(), lr=0.0, eps=1e-15)
        elif optimizer == "hip_sparse":
            from diff_gaussian_rasterization import SparseGaussianAdam
            self.optimizer = SparseGaussianAdam(model.param_groups(), lr=0.0, eps=1e-15)  # scene/gaussian_model.py:173
        else:
            raise ValueError(optimizer)
        self.loss_fn = {"hip": training_loss_fused, "torch": training_loss}[loss]
        self.bucket = GradBucket(model.parameters()) if world > 1 else None
        P = model.get_xyz.shape[0]
        self.xyz_gradient_accum = torch.zeros(P, 1, device=dev)
        self.denom = torch.zeros(P, 1, device=dev)
        self.max_radii2D = torch.zeros(P, device=dev)
        self.depth_targets, self.depth_weight = depth_targets, depth_weight
        # reference train.py:106 passes separate_sh=SPARSE_ADAM_AVAILABLE: dc / rest go to the rasterizer unconcatenated
        self.separate_sh = separate_sh
        self.last = {}

    def step(self, view_idx: int):
        cam = self.cameras[view_idx]
        pkg = self.render_fn(cam, self.model, self.pipe, self.bg, separate_sh=self.separate_sh)
        image, vsp, vis, radii = pkg["render"], pkg["viewspace_points"], pkg["visibility_filter"], pkg["radii"]
        loss = self.loss_fn(image, self.gt_images[view_idx], self.lambda_dssim)
        if self.depth_weight > 0 and self.depth_targets is not None:
            loss = loss + self.depth_weight * torch.abs(pkg["depth"] - self.depth_targets[view_idx]).mean()
        loss.backward()
        with torch.no_grad():
            # densification statistics are per-view: norm BEFORE any cross-rank reduction
            # same values as reference train.py:159-160 / gaussian_model.py:431-433, written without boolean-mask
            # indexing (which costs a device->host sync per step): radii and the gradient are 0 where not visible
            torch.maximum(self.max_radii2D, radii.float(), out=self.max_radii2D)
            self.xyz_gradient_accum += torch.norm(vsp.grad[:, :2], dim=-1, keepdim=True) * vis[:, None]
            self.denom += vis[:, None]
            if self.bucket is not None:
                self.bucket.all_reduce_mean(self.world)
            if self.optimizer_kind == "hip_sparse":
                self.optimizer.step(vis, radii.shape[0])                               # train.py:173-175
            else:
                self.optimizer.step()
            self.optimizer.zero_grad(set_to_none=True)
        self.last = dict(loss=loss.detach(), image=image.detach(), radii=radii)
        return self.last
