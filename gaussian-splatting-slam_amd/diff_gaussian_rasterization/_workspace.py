"""Grow-only state buffers for the rasterizer and the bookkeeping of the speculative forward.

The published rasterizer (and `gsr_forward_prepare`) reads `num_rendered` back to the host in the MIDDLE of every forward to
size its binning buffer: the device idles while the host wakes up, allocates and enqueues the rest, and every view gets a
differently sized `torch.empty`.  Here every in-flight forward -> backward pair leases one `Workspace` (geometry / binning /
image state + the backward's gradient records) from a per-device pool; buffers only ever grow, and the forward is enqueued
WHOLE through `gsr_forward_async`, sized by a capacity estimate for its shape (P, W, H).  Three modes:

  "exact" (default)  verified speculation: once the frame is enqueued the call waits for the frame's count, which the device
                     delivers to pinned memory as soon as it is known (the binning and compositing stages are still queued
                     behind it, so the device never idles), and if the count exceeds the capacity the binning state grows and
                     `gsr_forward_rerender` repeats phase 2.  EVERY frame is the blocking path's frame, bit for bit.
  "async"            no wait at all (a host-bound loop).  A frame beyond the capacity is composited from a truncated list; the
                     device-side guard (gsr_common.h, gsr_overflowed) makes its backward a no-op - zero gradients, no folded
                     optimizer step, no folded statistics - and the frame's ticket is reported by `take_overflowed()` as soon
                     as its status has arrived, so the caller can run that view again (scene_utils/trainer.py does).
  "sync"             the published blocking read-back (`gsr_forward_prepare` + `gsr_forward_render`).
"""
from __future__ import annotations

import collections
import os
import threading
import warnings

import torch

from . import _C

_MODES = ("exact", "async", "sync")
_MODE = os.environ.get("GSR_FORWARD_MODE", "exact").lower()
if _MODE not in _MODES:
    raise ValueError(f"GSR_FORWARD_MODE={_MODE!r}: expected one of {_MODES}")
# binning form of the speculative forward: "tile" (tile-local depth ordering, include/gsr.h gsr_forward_async) unless a
# shape's lists get too long for it, or "global" (GSR_BINNING=global: always the global depth sort of the blocking path)
_BINNING = os.environ.get("GSR_BINNING", "tile").lower()
TLO_MAX_LIST = 3072      # 75 % of the kernel's LDS capacity (4096 entries): beyond that a shape goes back to the global form
HEADROOM = float(os.environ.get("GSR_HEADROOM", "1.5"))    # capacity = max(num_rendered seen for this shape) * HEADROOM (1.25 -> 1.5 costs ~1 us per frame at C3)
MIN_CAPACITY = 1 << 14
FIRST_GUESS_PER_GAUSSIAN = 6      # capacity of a shape's first frame: 6 tile instances per Gaussian (C1-C5: 1.3 - 13; a low guess costs one re-render)


def set_forward_mode(mode: str):
    """"exact" (default): speculative forward, count verified before the call returns - every frame exact.  "async": no wait;
    an overflowed frame's backward is a no-op and its ticket is reported (take_overflowed).  "sync": always the blocking
    read-back of the published rasterizer."""
    global _MODE
    if mode not in _MODES:
        raise ValueError(mode)
    _MODE = mode


def forward_mode() -> str:
    return _MODE


def tile_local_binning(pool, key) -> bool:
    return _BINNING == "tile" and pool.longest_list.get(key, 0) <= TLO_MAX_LIST


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

    def __init__(self, pool, ws, recycle=True):
        self.pool, self.ws, self.recycle = pool, ws, recycle

    def release(self):
        ws, self.ws = self.ws, None
        if ws is not None and self.recycle:      # (a workspace captured into a HIP graph belongs to that graph for good)
            self.pool.give_back(ws)

    def __del__(self):
        try:
            self.release()
        except Exception:       # interpreter shutdown
            pass


WAIT_LIMIT_S = float(os.environ.get("GSR_COUNT_TIMEOUT_S", "120"))
STATUS_PENDING = -1          # third status word (longest tile list | 0) before the device has written the frame's status
STATUS_SORT_TIMEOUT = 1      # bit 0 of the first status word (csrc/gsr_common.h GSR_STATUS_SORT_TIMEOUT)


class _StatusArrived:
    """Event-like view of a pinned status slot: "done" once the device has overwritten the sentinel in its third word (the low
    half: an aligned 4-byte store, never seen torn)."""

    def __init__(self, status):
        self.status = status

    def query(self):
        return (int(self.status[2]) & 0xFFFFFFFF) != 0xFFFFFFFF


def _wait_bounded(ev, what):
    """`ev.synchronize()` with a bound: no host wait of this module may sit forever behind a device that stopped making progress
    (the native side bounds its own wait for the count the same way, api.hip wait_for_count)."""
    import time
    t0 = time.monotonic()
    spins = 0
    while not ev.query():
        spins += 1
        if spins > 2000:
            time.sleep(50e-6)
            if time.monotonic() - t0 > WAIT_LIMIT_S:
                raise _C.GsrError(f"gsr: {what} did not arrive within {WAIT_LIMIT_S:.0f} s (device hung?)")


class Pool:
    def __init__(self, device):
        self.device = device
        self.lock = threading.Lock()
        self.free = []
        self.capacity = {}                       # (P, W, H) -> instances the binning state is sized for
        self.longest_list = {}                   # (P, W, H) -> longest tile list seen (tile-local binning form)
        self.pending = collections.deque()       # (event, pinned status, capacity used, key, ticket, count already verified)
        self.status_free = []
        self.ticket = 0                          # frames handed to gsr_forward_async so far
        self.overflowed = []                     # tickets of unverified frames that turned out to be truncated
        self.stats = {"num_rendered": 0, "overflow_frames": 0, "async_frames": 0, "sync_frames": 0, "exact_frames": 0,
                      "rerendered_frames": 0}

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

    def capacity_for_frame(self, key):
        """Capacity the next frame of this shape is enqueued with: what the shape has shown so far (plus headroom), or a guess
        for its first frame."""
        cap = self.capacity.get(key)
        return cap if cap is not None else max(MIN_CAPACITY, (FIRST_GUESS_PER_GAUSSIAN * key[0] + 4095) & ~4095)

    def note(self, key, R):
        self.stats["num_rendered"] = int(R)
        cap = self.capacity.get(key)
        if cap is None or R * 1.1 > cap:
            self.capacity[key] = _capacity_for(R)

    def forget_estimates(self):
        """Drops what was learnt per shape (capacities, list lengths)."""
        self.poll(wait=True)
        self.capacity.clear(); self.longest_list.clear()
        self.overflowed.clear()

    def poll(self, wait=False):
        """Looks at the statuses of earlier speculative forwards that have completed (all of them with wait=True)."""
        while self.pending:
            ev, status, cap, key, ticket, verified = self.pending[0]
            if ev is None:
                ev = _StatusArrived(status)
            if wait:
                _wait_bounded(ev, f"the status of frame {ticket}")
            elif not ev.query():
                break
            self.pending.popleft()
            flags, R = int(status[0]), int(status[1])
            longest = int(status[2]) & 0xFFFFFFFF
            self.longest_list[key] = max(self.longest_list.get(key, 0), longest)
            self.status_free.append(status)
            if flags & STATUS_SORT_TIMEOUT:
                # (reference failure contract, README.md:168-171: a failing rasterizer call raises.  Without debug the kernels
                # are fire-and-forget, so the report arrives with the frame's status, one call later.)
                raise _C.GsrError(f"gsr: frame {ticket} (P, W, H = {key}): a radix-sort look-back wait timed out on the device "
                                  "(inter-workgroup hand-off broken) - its tile lists, image and gradients are not to be trusted")
            if verified:        # "exact" mode: the count was looked at (and acted upon) before the forward returned
                continue
            if int(status[3]) & 0xFFFFFFFF:
                # a tile list truncated by its depth cut-off was too short (gsr_forward_async_culled): the frame flagged itself,
                # its backward was a no-op; the caller renders the view again (untruncated).  Nothing to learn about capacities.
                self.stats["cull_miss_frames"] = self.stats.get("cull_miss_frames", 0) + 1
                self.overflowed.append(ticket)
                continue
            if (flags >> 32) & 1:
                raise _C.GsrError("Point is filtered although prefiltered is set. This shouldn't happen!")
            if R > cap or R < 0:
                self.stats["overflow_frames"] += 1
                self.overflowed.append(ticket)
                warnings.warn(f"gsr: frame {ticket} (P, W, H = {key}) had {R} tile instances, the binning state held {cap}: it was "
                              "composited from a truncated list and its backward was a no-op; the capacity has been raised "
                              "(forward mode 'exact', the default, re-renders such a frame instead)", RuntimeWarning)
            self.note(key, R if R >= 0 else 0x7FFFFFFF // 2)

    def take_overflowed(self):
        """Tickets (see `last_ticket`) of unverified frames found truncated since the last call."""
        out, self.overflowed = self.overflowed, []
        return out


_pools = {}


def pool(device) -> Pool:
    idx = device.index if device.index is not None else torch.cuda.current_device()
    p = _pools.get(idx)
    if p is None:
        p = _pools[idx] = Pool(torch.device("cuda", idx))
    return p
