"""Multi-rank path on the CPU (gloo, world size 2): reach sharding, per-shard parameter seeding and
the hydrograph all_gather reproduce the single-process result in global reach order.  The stepping
itself is done by the CPU oracle here (tiny reaches) - on the GPU box the same shard/gather code
runs around the HIP kernel with backend "nccl" (bench.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from flowsim_amd.shard import gather_hydrographs, gather_hydrographs_split, reach_block, split_reaches
from flowsim_amd.synthetic import c3_reach_parameters, inflow_table, normal_depth_rect

N, STEPS, PER_RANK, DT, DX = 12, 3, 3, 600.0, 250.0


def hydrographs_of_block(first, count):
    """[levels, 4, count] for global reaches [first, first+count) (oracle as the stand-in stepper)."""
    from oracle import preissmann_oracle as O
    b, n, S0, Qb = c3_reach_parameters(first, count)
    hn = normal_depth_rect(b, n, S0, Qb)
    tgt = inflow_table(Qb, STEPS + 1, DT)
    out = np.empty((STEPS + 1, 4, count))
    for r in range(count):
        geo = {k: np.zeros(N) for k in O.GEO_KEYS}
        geo["b_main"][:] = b[r]; geo["n_main"][:] = n[r]; geo["n_left"][:] = n[r]; geo["n_right"][:] = n[r]
        L = (N - 1) * DX
        geo["z_bed"] = S0[r] * L * (1 - np.arange(N) / (N - 1))
        p = O.Problem(geo=geo, h0=np.full(N, hn[r]), Q0=np.full(N, Qb[r]),
                      us=O.BC("flow_hydrograph", bed_level=S0[r] * L, target=tgt[:, r].copy()),
                      ds=O.BC("normal_depth", bed_level=0.0, bed_slope=float(S0[r])),
                      theta=0.6, dt=DT, dx=DX, nt=STEPS + 1, tol=1e-6)
        res = O.newton_run(p)
        out[:, 0, r], out[:, 1, r] = res["depth"][:, 0], res["flow"][:, 0]
        out[:, 2, r], out[:, 3, r] = res["depth"][:, -1], res["flow"][:, -1]
    return out


TOTAL_STRONG = 7      # strong scaling: 7 reaches over 2 ranks = blocks of 4 and 3


def _worker(rank, world, port, q, strong=False, total_strong=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if strong:      # bench.py --total-reaches: contiguous blocks of unequal size, the shorter one travels padded
        total = total_strong or TOTAL_STRONG
        first, count = split_reaches(total, rank, world)
        local = torch.from_numpy(hydrographs_of_block(first, count))
        allh = gather_hydrographs_split(local, total, world)
        root0 = gather_hydrographs_split(local, total, world, 0)          # bench.py's form: to rank 0 alone
    else:
        first, count = reach_block(rank, world, PER_RANK)
        local = torch.from_numpy(hydrographs_of_block(first, count))
        allh = gather_hydrographs(local, world)
        root0 = gather_hydrographs(local, world, 0)
    dist.barrier()
    assert (root0 is None) == (rank != 0)
    if rank == 0:
        assert torch.equal(root0, allh)
        q.put(root0.numpy())
    dist.destroy_process_group()


def test_two_rank_gather_equals_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = hydrographs_of_block(0, 2 * PER_RANK)
    assert got.shape == want.shape == (STEPS + 1, 4, 2 * PER_RANK)
    assert np.array_equal(got, want)            # shard-invariant, bit for bit


def test_two_rank_strong_scaling_gather_equals_single_process():
    """the --total-reaches path of bench.py: split_reaches + gather_hydrographs_split at world size 2"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, True)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = hydrographs_of_block(0, TOTAL_STRONG)
    assert got.shape == want.shape == (STEPS + 1, 4, TOTAL_STRONG)
    assert np.array_equal(got, want)


def test_parameter_stream_is_shard_invariant():
    whole = c3_reach_parameters(0, 10)
    for first, count in ((0, 4), (4, 3), (7, 3)):
        part = c3_reach_parameters(first, count)
        for a, b in zip(whole, part):
            assert np.array_equal(a[first:first + count], b)


def test_split_reaches_covers_everything():
    for total, world in ((10, 3), (8, 8), (65536, 8), (5, 2)):
        seen = []
        for r in range(world):
            f, c = split_reaches(total, r, world)
            seen += list(range(f, f + c))
        assert seen == list(range(total))


@pytest.mark.parametrize("strong", [False, True])
def test_eight_ranks_as_the_scaling_run_has_them(strong):
    """The driver's scaling tier goes to --gpus 8; the build box has one GPU, so the eight-rank control flow is rehearsed here on the CPU
    (gloo): eight processes, rank r its own block of reaches (weak: 3 each; strong: 29 reaches in blocks of 4 and 3, the shorter ones
    padded), the gather to rank 0 and the all-ranks form, against one process stepping all the reaches - bit for bit."""
    total = 29
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 8, port, q, strong, total)) for r in range(8)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = hydrographs_of_block(0, total if strong else 8 * PER_RANK)
    assert got.shape == want.shape and np.array_equal(got, want)
