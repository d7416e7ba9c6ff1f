"""The overlapped data-parallel schedule (SH exchange + Adam on a side stream under the next step's geometry stages)
gives bit-identical parameters to the plain synchronous schedule.  Two ranks share the one GPU of the test box (gloo between
them: RCCL needs one device per rank), so this checks the schedule's correctness, not its speed."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_overlapped_schedule_is_bit_identical():
    port = 29600 + (os.getpid() % 300)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dp_overlap_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "DP_OVERLAP_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_every_exchange_runs_on_rccl_with_one_rank():
    """The collectives of the N > 1 schedule on REAL RCCL (backend "nccl"; one rank, because RCCL wants one device per rank and
    the box has one): all-reduce AVG in place on the six leaf gradients, uint8 MAX, all_gather_into_tensor (also in place on a
    slice of its own output), reduce_scatter_tensor, async work issued from a side stream - each schedule must give the plain
    single-GPU run's parameters (tests/rccl_single_rank_worker.py)."""
    cmd = [sys.executable, os.path.join(ROOT, "tests", "rccl_single_rank_worker.py")]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0 and "RCCL_SINGLE_RANK_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize("exchange", ["allreduce", "sharded", "sh_rank1"])
def test_bench_gpus_2_starts_two_ranks_itself(exchange):
    """`python bench.py --gpus 2` with no launcher around it (how the driver invokes --gpus 1) must start two ranks itself and
    report n_gpus = 2 (VERDICT r1 / ADVICE: it used to run ONE rank and print a 1-GPU number).  Two ranks share this box's
    one card over gloo (BENCH_SHARE_GPU / BENCH_BACKEND: RCCL needs one device per rank), small config, few steps."""
    import json
    env = dict(os.environ, BENCH_BACKEND="gloo", BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--config", "1",
           "--views", "4", "--no-cpu-baseline", "--no-kernel-profile", "--optimizer", "hip", "--exchange", exchange]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["config"]["exchange"] == exchange
    # and a launcher environment of the wrong size is refused instead of mislabelled
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r2 = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env2)
    assert r2.returncode != 0 and "refusing" in r2.stderr


@pytest.mark.timeout(900)
@pytest.mark.parametrize("exchange", ["sh_rank1", "allreduce", "sharded"])
def test_bench_launches_three_ranks_on_the_shared_card(exchange):
    """More than two ranks on the test box.  Its process guard admits SIX processes with the card open, and this pytest process
    and bench.py's launching parent are two of them (a 5-rank attempt was killed by the guard: 7 processes, gpurun_out/r4b) - so
    3 ranks here; north_star's 8 ranks run under gloo on the CPU (tests/test_distributed_cpu.py::
    test_eight_ranks_every_exchange_equals_the_mean_gradient_step).  `python bench.py --gpus 3` with no launcher: starts the ranks
    itself, 10 000 Gaussians (10 000 mod 3 = 1: ShardedStep's left-over row) and 7 views (7 mod 3 = 1: ranks get 3 / 2 / 2),
    overlapped exchange where the schedule has one (the default at N > 1), gloo between the ranks.  Asserts rc 0, n_gpus 3 and the
    exchange figure of the JSON line."""
    import json
    env = dict(os.environ, BENCH_BACKEND="gloo", BENCH_SHARE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "3", "--warmup", "1", "--config", "1",
           "--views", "7", "--no-cpu-baseline", "--no-kernel-profile", "--optimizer", "hip", "--exchange", exchange]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=800, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    # the contract: ONE JSON line on stdout, whatever the ranks' libraries print (bench.py points descriptor 1 at stderr; round 4:
    # librccl writes its version banner to stdout, and a rank that dies in the teardown used to take a buffered line with it)
    assert len(r.stdout.strip().splitlines()) == 1, r.stdout[-2000:]
    d = json.loads(r.stdout)
    assert d["n_gpus"] == 3 and d["steps"] == 3 and d["value"] > 0 and d["config"]["exchange"] == exchange
    row = 4 * (11 + 3 * 1)                                                                       # SH degree 0 at C1: 14 floats
    from_table = {"sh_rank1": 2 * 2 / 3 * 44 + 2 * 12, "allreduce": 2 * 2 / 3 * row, "sharded": 2 * 2 / 3 * row}[exchange]
    assert abs(d["config"]["exchange_bytes_per_gaussian_received"] - from_table) < 0.1


def test_bench_prints_one_line_beside_rccl():
    """`BENCH_SINGLE_RANK_GROUP=1 python bench.py --gpus 1`: the whole N > 1 schedule over a one-rank RCCL group.  librccl prints
    a version banner on descriptor 1 when the communicator comes up; the bench line must still be the ONLY line of stdout (the
    driver parses it), and it must be there even though the process goes through RCCL's teardown after printing it."""
    import json
    env = dict(os.environ, BENCH_SINGLE_RANK_GROUP="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_BACKEND", "BENCH_SHARE_GPU"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--config", "1",
           "--views", "4", "--no-cpu-baseline", "--no-kernel-profile"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert len(r.stdout.strip().splitlines()) == 1, r.stdout[-2000:]
    d = json.loads(r.stdout)
    assert d["n_gpus"] == 1 and d["config"].get("single_rank_group") is True and d["value"] > 0


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_rank1_expand_kernel_against_torch(deg):
    """csrc/exchange.hip against the same arithmetic in torch ops (scene_utils.parallel's CPU branch): three "ranks", a third of
    the per-rank gradients exact zeros (Gaussians without instances in that rank's view); |err| <= 1e-6 max(1, |value|)."""
    import ctypes as C
    import torch
    sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-slam_amd"))
    from diff_gaussian_rasterization import _C
    from scene_utils.sh import sh_basis, C0
    gen = torch.Generator().manual_seed(77 + deg)
    P, N, krest = 1000, 3, (3 + 1) ** 2 - 1
    xyz = torch.randn(P, 3, generator=gen)
    gathered = torch.zeros(N, P + 1, 3)
    for r in range(N):
        g = torch.randn(P, 3, generator=gen)
        g[torch.rand(P, generator=gen) < 0.33] = 0.0
        gathered[r, :P] = g
        gathered[r, P] = 4.0 * torch.nn.functional.normalize(torch.randn(3, generator=gen), dim=0)
    K = (deg + 1) ** 2
    acc_dc = torch.zeros(P, 3, dtype=torch.float64)
    acc = torch.zeros(P, krest, 3, dtype=torch.float64)
    for r in range(N):
        gr, cam = gathered[r, :P].double(), gathered[r, P].double()
        acc_dc += gr
        if K > 1:
            d = xyz.double() - cam
            d = d / d.norm(dim=1, keepdim=True)
            w = sh_basis(deg, d)[:, 1:K] / C0
            acc[:, :K - 1] += w[:, :, None] * gr[:, None, :]
    exp_dc, exp_rest = acc_dc / N, acc / N
    dev = "cuda"
    out_dc = torch.full((P, 1, 3), float("nan"), device=dev)
    out_rest = torch.full((P, krest, 3), float("nan"), device=dev)
    xd, gd = xyz.to(dev), gathered.to(dev).contiguous()
    _C.check(_C.lib().gsr_sh_rank1_expand(P, N, deg, krest, _C.ptr(xd), _C.ptr(gd), C.c_float(1.0 / N), _C.ptr(out_dc),
                                          _C.ptr(out_rest), _C._stream()))
    torch.cuda.synchronize()
    for got, exp in ((out_dc.cpu().view(P, 3), exp_dc), (out_rest.cpu(), exp_rest)):
        err = (got.double() - exp).abs()
        assert bool((err <= 1e-6 * exp.abs().clamp(min=1.0)).all()), float(err.max())
    assert not out_rest[:, K - 1:].any() if K - 1 < krest else True       # coefficients beyond the active degree: zeros


def test_sh_rank1_adam_kernel_is_expand_plus_adam_bit_for_bit():
    """gsr_sh_rank1_adam (rebuild + dense Adam of f_dc / f_rest in one kernel) against gsr_sh_rank1_expand followed by
    FusedAdam.step on the two tensors, three steps (so the moments matter): parameters and both moments bit for bit."""
    import ctypes as C
    import torch
    sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-slam_amd"))
    from diff_gaussian_rasterization import _C, FusedAdam
    from scene_utils.parallel import _fused_sh_adam_args
    dev, P, N, deg, krest = "cuda", 3001, 4, 3, 15
    gen = torch.Generator().manual_seed(91)
    xyz = torch.randn(P, 3, generator=gen).to(dev)

    def make():
        g2 = torch.Generator().manual_seed(92)
        f_dc = torch.nn.Parameter(torch.randn(P, 1, 3, generator=g2).to(dev))
        f_rest = torch.nn.Parameter(torch.randn(P, krest, 3, generator=g2).to(dev))
        opt = FusedAdam([{"params": [f_dc], "lr": 0.0025, "name": "f_dc"}, {"params": [f_rest], "lr": 0.000125, "name": "f_rest"}],
                        lr=0.0, eps=1e-15)
        return f_dc, f_rest, opt
    a_dc, a_rest, a_opt = make()
    b_dc, b_rest, b_opt = make()
    lib = _C.lib()
    for it in range(3):
        gathered = torch.randn(N, P + 1, 3, generator=gen)
        gathered[:, :P][torch.rand(N, P, generator=gen) < 0.3] = 0.0
        gathered = gathered.to(dev).contiguous()
        # (a) expand -> .grad -> FusedAdam.step
        a_dc.grad, a_rest.grad = torch.empty_like(a_dc), torch.empty_like(a_rest)
        _C.check(lib.gsr_sh_rank1_expand(P, N, deg, krest, _C.ptr(xyz), _C.ptr(gathered), C.c_float(1.0 / N), _C.ptr(a_dc.grad),
                                         _C.ptr(a_rest.grad), _C._stream()))
        a_opt.step()
        # (b) one kernel
        fa, keep = _fused_sh_adam_args(b_opt, b_dc, b_rest)
        _C.check(lib.gsr_sh_rank1_adam(P, N, deg, krest, _C.ptr(xyz), _C.ptr(gathered), C.c_float(1.0 / N), _C.ptr(b_dc.data),
                                       _C.ptr(b_rest.data), C.byref(fa), _C._stream()))
    torch.cuda.synchronize()
    for x, y, ox, oy in ((a_dc, b_dc, a_opt, b_opt), (a_rest, b_rest, a_opt, b_opt)):
        assert torch.equal(x.data, y.data)
        assert torch.equal(ox.state[x]["exp_avg"], oy.state[y]["exp_avg"])
        assert torch.equal(ox.state[x]["exp_avg_sq"], oy.state[y]["exp_avg_sq"])
        assert int(ox.state[x]["step"]) == int(oy.state[y]["step"]) == 3
