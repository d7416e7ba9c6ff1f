"""Grow-only state buffers for the rasterizer and the bookkeeping of the non-blocking forward.

The published rasterizer (and `gsr_forward_prepare`) reads `num_rendered` back to the host in every forward to size its
binning buffer: one blocking wait per frame, and a differently sized `torch.empty` per view, which keeps the caching
allocator growing for a whole pass over the views.  Here every in-flight forward -> backward pair leases one `Workspace`
(geometry / binning / image state + the backward's gradient records) from a per-device pool; buffers only ever grow
(geometrically), and once the instance count of a shape (P, W, H) is known the forward goes through `gsr_forward_async`:
no wait, `num_rendered` arrives in pinned memory and is looked at by a LATER call (`Pool.poll`), where it raises the capacity
estimate for the next frames.  A frame whose count exceeds the estimate loses instances (the farthest ones in the global binning
form, arbitrary ones in the tile-local form, which is why that form waits for a settled capacity; include/gsr.h,
gsr_forward_async) and is counted in `stats["overflow_frames"]`; `set_forward_mode("sync")` / GSR_FORWARD_MODE=sync keeps the
blocking read-back for callers that cannot accept that.
"""
from __future__ import annotations

import collections
import os
import threading
import warnings

import torch

from . import _C

_MODE = os.environ.get("GSR_FORWARD_MODE", "async").lower()
# binning form of the non-blocking forward: "tile" (tile-local depth ordering, include/gsr.h gsr_forward_async) unless a
# shape's lists get too long for it, or "global" (GSR_BINNING=global: always the global depth sort of the blocking path)
_BINNING = os.environ.get("GSR_BINNING", "tile").lower()
TLO_MAX_LIST = 3072      # 75 % of the kernel's LDS capacity (4096 entries): beyond that a shape goes back to the global form
# The tile-local form emits in index order, so a frame beyond the capacity loses ARBITRARY instances where the global form
# loses the farthest ones (usually invisible).  It is therefore used only once a shape's capacity has held for this many
# consecutive frames (GSR_TLO_SETTLE); any raise of the capacity starts the count again.
TLO_SETTLE_FRAMES = int(os.environ.get("GSR_TLO_SETTLE", "3"))
HEADROOM = float(os.environ.get("GSR_HEADROOM", "1.5"))    # capacity = max(num_rendered seen for this shape) * HEADROOM (1.25 -> 1.5 costs ~1 us per frame at C3)
MIN_CAPACITY = 1 << 14


def set_forward_mode(mode: str):
    """"async" (default): non-blocking forward once a shape's instance count is known.  "sync": always the blocking
    read-back of the published rasterizer (exact for every frame, the host stays at most one frame ahead)."""
    global _MODE
    if mode not in ("async", "sync"):
        raise ValueError(mode)
    _MODE = mode


def forward_mode() -> str:
    return _MODE


def tile_local_binning(pool, key) -> bool:
    return (_BINNING == "tile" and pool.longest_list.get(key, 0) <= TLO_MAX_LIST
            and pool.settled.get(key, 0) >= TLO_SETTLE_FRAMES)


def _capacity_for(R: int) -> int:
    return max(MIN_CAPACITY, (int(R * HEADROOM) + 4095) & ~4095)


class Workspace:
    """State buffers of ONE forward -> backward pair (opaque to Python; layouts live in csrc/gsr_common.h)."""

    def __init__(self, device):
        self.device = device
        self.geom = self.img = self.binning = self.scratch = None
        self.stream = None            # raw hipStream_t (int) the last user enqueued on

    def _grow(self, name, nbytes, slack=1.0):
        t = getattr(self, name)
        if t is None or t.numel() < nbytes:
            t = torch.empty(int(nbytes * slack) + 256, dtype=torch.uint8, device=self.device)
            setattr(self, name, t)
        return t

    def ensure_geom(self, lib, P):
        return self._grow("geom", lib.gsr_geometry_state_bytes(P), 1.1 if self.geom is not None else 1.0)

    def ensure_img(self, lib, W, H):
        return self._grow("img", lib.gsr_image_state_bytes(W, H))

    def ensure_binning(self, lib, P, W, H, cap):
        return self._grow("binning", lib.gsr_binning_state_bytes(P, W, H, cap))

    def ensure_scratch(self, lib, P, cap):
        return self._grow("scratch", lib.gsr_backward_scratch_bytes(P, cap))


class Lease:
    """Held by the autograd ctx; the workspace goes back to the pool when the ctx dies (after backward, or when the
    outputs are dropped without one)."""

    def __init__(self, pool, ws):
        self.pool, self.ws = pool, ws

    def release(self):
        ws, self.ws = self.ws, None
        if ws is not None:
            self.pool.give_back(ws)

    def __del__(self):
        try:
            self.release()
        except Exception:       # interpreter shutdown
            pass


class Pool:
    def __init__(self, device):
        self.device = device
        self.lock = threading.Lock()
        self.free = []
        self.capacity = {}                       # (P, W, H) -> instances the binning state is sized for
        self.longest_list = {}                   # (P, W, H) -> longest tile list seen (tile-local binning form)
        self.settled = {}                        # (P, W, H) -> frames noted since the capacity last changed
        self.pending = collections.deque()       # (event, pinned status, capacity used, key)
        self.status_free = []
        self.stats = {"num_rendered": 0, "overflow_frames": 0, "async_frames": 0, "sync_frames": 0}
        self._warned = False

    # ---- workspaces ----
    def acquire(self) -> Workspace:
        stream = _C.raw_stream(self.device.index)
        with self.lock:
            ws = None
            for i, w in enumerate(self.free):     # prefer one last used on this stream (plain stream order protects it)
                if w.stream is None or w.stream == stream:
                    ws = self.free.pop(i)
                    break
            if ws is None and self.free:
                ws = self.free.pop()
        if ws is None:
            ws = Workspace(self.device)
        elif ws.stream is not None and ws.stream != stream:
            # its previous user may still be running on another stream
            torch.cuda.current_stream(self.device).wait_stream(torch.cuda.ExternalStream(ws.stream))
        ws.stream = stream
        return ws

    def give_back(self, ws):
        with self.lock:
            self.free.append(ws)

    # ---- num_rendered bookkeeping ----
    def status_slot(self):
        if self.status_free:
            return self.status_free.pop()
        return torch.zeros(4, dtype=torch.int64).pin_memory()    # [flags word 0|1, num_rendered, longest list|-, -]

    def note(self, key, R):
        self.stats["num_rendered"] = int(R)
        cap = self.capacity.get(key)
        if cap is None or R * 1.1 > cap:
            self.capacity[key] = _capacity_for(R)
            self.settled[key] = 0
        else:
            self.settled[key] = self.settled.get(key, 0) + 1

    def forget_estimates(self):
        """Drops what was learnt per shape (capacities, list lengths): the next frame of every shape blocks once again."""
        self.poll(wait=True)
        self.capacity.clear(); self.longest_list.clear(); self.settled.clear()

    def poll(self, wait=False):
        """Looks at the statuses of earlier non-blocking forwards that have completed (all of them with wait=True)."""
        while self.pending:
            ev, status, cap, key = self.pending[0]
            if wait:
                ev.synchronize()
            elif not ev.query():
                break
            self.pending.popleft()
            flags, R = int(status[0]), int(status[1])
            longest = int(status[2]) & 0xFFFFFFFF
            self.longest_list[key] = max(self.longest_list.get(key, 0), longest)
            self.status_free.append(status)
            if (flags >> 32) & 1:
                raise _C.GsrError("Point is filtered although prefiltered is set. This shouldn't happen!")
            if R > cap or R < 0:
                self.stats["overflow_frames"] += 1
                if not self._warned:
                    self._warned = True
                    warnings.warn(f"gsr: a frame had {R} tile instances, capacity was {cap}: the instances beyond it "
                                  "(the farthest ones, or arbitrary ones in the tile-local binning form) were dropped "
                                  "for that frame; the capacity has been raised (GSR_FORWARD_MODE=sync avoids this)")
            self.note(key, R if R >= 0 else 0x7FFFFFFF // 2)


_pools = {}


def pool(device) -> Pool:
    idx = device.index if device.index is not None else torch.cuda.current_device()
    p = _pools.get(idx)
    if p is None:
        p = _pools[idx] = Pool(torch.device("cuda", idx))
    return p
