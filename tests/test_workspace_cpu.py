"""Host-side bookkeeping of the speculative forward (diff_gaussian_rasterization/_workspace.py) without a GPU: capacity
estimates, the first-frame guess, per-frame overflow reports of the unverified mode, mode switching.  Events and pinned status
slots are stood in for by plain objects; no library call is made."""
import warnings

import pytest
import torch


class _Ev:
    """done=False: completes at the fourth look (a non-blocking poll sees it pending, a waiting one gets there)"""

    def __init__(self, done=True):
        self.done, self.looks = done, 0

    def query(self):
        self.looks += 1
        return self.done or self.looks > 3


def _pool():
    from diff_gaussian_rasterization import _workspace as ws
    p = ws.Pool(torch.device("cpu"))
    p.status_slot = lambda: torch.zeros(4, dtype=torch.int64)          # (the real one pins the memory)
    return ws, p


def test_modes_and_default():
    import os
    from diff_gaussian_rasterization import _workspace as ws
    old = ws.forward_mode()
    try:
        if "GSR_FORWARD_MODE" not in os.environ:
            assert old == "exact"
        for m in ("exact", "async", "sync"):
            ws.set_forward_mode(m)
            assert ws.forward_mode() == m
        with pytest.raises(ValueError):
            ws.set_forward_mode("fast")
    finally:
        ws.set_forward_mode(old)


def test_first_frame_guess_then_measured_capacity_with_headroom():
    ws, p = _pool()
    key = (100_000, 1920, 1080)
    guess = p.capacity_for_frame(key)
    assert guess >= ws.FIRST_GUESS_PER_GAUSSIAN * key[0] and guess % 4096 == 0 and key not in p.capacity
    p.note(key, 1_340_000)
    cap = p.capacity[key]
    assert cap >= int(1_340_000 * ws.HEADROOM) and cap % 4096 == 0 and p.capacity_for_frame(key) == cap
    p.note(key, 1_400_000)                      # within the headroom: the estimate holds
    assert p.capacity[key] == cap
    p.note(key, int(cap / 1.05))                # closer than 10 % to the capacity: raised
    assert p.capacity[key] > cap
    assert p.capacity_for_frame((3, 16, 16)) == ws.MIN_CAPACITY


def test_unverified_frames_are_reported_per_ticket_and_verified_ones_are_not():
    ws, p = _pool()
    key = (1000, 64, 64)
    p.capacity[key] = 20000

    def frame(R, ticket, verified, done=True, longest=0):
        st = p.status_slot()
        st[0], st[1], st[2] = 0, R, longest
        p.pending.append((_Ev(done), st, 20000, key, ticket, verified))
    frame(15000, 1, False)
    frame(50000, 2, False)                      # truncated
    frame(70000, 3, True)                       # verified by the forward itself (it re-rendered): not an overflow
    frame(90000, 4, False, done=False)          # status not there yet
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        p.poll()
    assert [str(x.message).split()[2] for x in w] == ["2"]           # "gsr: frame 2 ..."
    assert p.stats["overflow_frames"] == 1 and p.take_overflowed() == [2] and p.take_overflowed() == []
    assert len(p.pending) == 1 and p.capacity[key] >= 50000
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        p.poll(wait=True)
    assert p.take_overflowed() == [4] and p.stats["overflow_frames"] == 2 and not p.pending
    assert p.stats["num_rendered"] == 90000


def test_long_lists_send_a_shape_back_to_the_global_binning_form():
    ws, p = _pool()
    key = (9000, 96, 64)
    if ws._BINNING != "tile":
        pytest.skip("GSR_BINNING=global")
    assert ws.tile_local_binning(p, key)
    st = p.status_slot()
    st[1], st[2] = 5000, 5000
    p.capacity[key] = 16384
    p.pending.append((_Ev(), st, 16384, key, 1, True))
    p.poll()
    assert p.longest_list[key] == 5000 and not ws.tile_local_binning(p, key)
    p.forget_estimates()
    assert ws.tile_local_binning(p, key) and not p.capacity


def test_verified_frames_need_no_event_the_status_sentinel_tells():
    """A verified frame is queued without an event: its status counts as arrived once the device has overwritten the sentinel in
    the low half of the third word (the longest tile list); frames behind a pending one wait their turn."""
    ws, p = _pool()
    key = (9000, 96, 64)
    p.capacity[key] = 16384
    a, b = p.status_slot(), p.status_slot()
    a[2] = b[2] = ws.STATUS_PENDING
    p.pending.append((None, a, 16384, key, 1, True))
    p.pending.append((None, b, 16384, key, 2, True))
    p.poll()
    assert len(p.pending) == 2 and key not in p.longest_list
    b[1], b[2] = 4000, 3000                      # the later frame's status is there, the earlier one's is not: order is kept
    p.poll()
    assert len(p.pending) == 2
    a[1], a[2] = 4000, -(1 << 32) + 1200          # high half still the sentinel's (0xFFFFFFFF): only the low half counts
    p.poll()
    assert not p.pending and p.longest_list[key] == 3000 and len(p.status_free) == 2


def test_dc_gradient_tail_row_is_recognised_only_for_the_backwards_own_tensor():
    """The backward returns its gradients in one arena with three spare floats behind each; `dc_grad_tail_row` hands out the
    [P + 1, 3] view over a dc gradient and its spare row only while the tensor still IS that allocation's - the knowledge rides on
    the arena's storage object (no process-wide "most recent backward"): it survives autograd moving the tensor into `.grad`, and a
    copy, a sum accumulated elsewhere, another backward's arena or a tensor that is no arena member gets None (the caller then
    concatenates)."""
    import diff_gaussian_rasterization as dgr
    P = 37
    parts = (((P, 3), 0), ((P, 1), 0), ((P, 3), 0), ((P, 4), 0), ((P, 1, 3), 3), ((P, 15, 3), 0), (None, 0), (None, 0))
    xyz, op, sc, rot, dc, rest, _, _ = dgr._grad_arena("cpu", parts)
    dc.copy_(torch.arange(3 * P, dtype=torch.float32).view(P, 1, 3))
    rest.fill_(-5.0)
    tail = dgr.dc_grad_tail_row(dc)
    assert tail is not None and tail.shape == (P + 1, 3) and tail.data_ptr() == dc.data_ptr()
    assert torch.equal(tail[:P], dc.view(P, 3))
    tail[P] = torch.tensor([7.0, 8.0, 9.0])
    assert bool((rest == -5.0).all())                      # the spare row lies in front of the next tensor
    assert dgr.dc_grad_tail_row(dc.clone()) is None
    assert dgr.dc_grad_tail_row(torch.zeros(P, 1, 3)) is None
    assert dgr.dc_grad_tail_row(dc.view(-1)[3:].view(P - 1, 1, 3)) is None          # same storage, not a member of the arena
    # two arenas alive at once (two devices' / two threads' / two interleaved backwards): each answers for its own tensor
    other = dgr._grad_arena("cpu", parts)[4]
    assert dgr.dc_grad_tail_row(other).data_ptr() == other.data_ptr() and dgr.dc_grad_tail_row(dc).data_ptr() == dc.data_ptr()

    # through autograd: a gradient the engine moves into `.grad` keeps its arena; an accumulated one does not
    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 2.0

        @staticmethod
        def backward(ctx, g):
            out = dgr._grad_arena("cpu", (((P, 1, 3), 3),))[0]
            out.fill_(1.0)
            return out
    a = torch.nn.Parameter(torch.zeros(P, 1, 3))
    Fn.apply(a).sum().backward()
    assert dgr.dc_grad_tail_row(a.grad) is not None and dgr.dc_grad_tail_row(a.grad).shape == (P + 1, 3)
    b = torch.nn.Parameter(torch.zeros(P, 1, 3))
    (Fn.apply(b) + Fn.apply(b)).sum().backward()
    assert dgr.dc_grad_tail_row(b.grad) is None


def test_no_process_wide_hand_offs_are_left_in_the_rasterizer_module():
    """no global state, re-entrant per device.  The optimizer fold, the statistics fold, the skipped dL/dsh_rest, the
    SH-ready event and the forward mode of a call travel as arguments of that call (BackwardFold on its autograd ctx)."""
    import inspect
    import diff_gaussian_rasterization as dgr
    for name in ("_fused_optimizer", "_fold_stats", "_skip_sh_rest", "_sh_ready_event", "_last_dc_grad", "_split_rows",
                 "fuse_optimizer_into_next_backward", "fold_densification_stats_into_next_backward",
                 "skip_sh_rest_grad_in_next_backward", "defer_sh_until"):
        assert not hasattr(dgr, name), name
    sig = inspect.signature(dgr.GaussianRasterizer.forward)
    for kw in ("fold", "sh_ready_event", "forward_mode"):
        assert sig.parameters[kw].kind is inspect.Parameter.KEYWORD_ONLY and sig.parameters[kw].default is None
    f = dgr.BackwardFold(optimizer=None, stats=(None, None, None))
    assert f.stats is None and not (f.optimizer_taken or f.stats_taken or f.sh_rest_skipped)
