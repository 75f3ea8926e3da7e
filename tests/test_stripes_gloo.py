"""The multi-rank path on CPU: two gloo ranks each fill their own stripes of a frame, gather, and
must both end with the full frame.  Covers the partition rule rt_render uses (stripe s -> rank s % world)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd.stripes import StripePlan  # noqa: E402


def reference_frame(H, W):
    """Screen-order frame whose value encodes (image row, column, channel)."""
    y = torch.arange(H - 1, -1, -1, dtype=torch.float32).view(H, 1, 1)  # memory row m = image row H-1-m
    x = torch.arange(W, dtype=torch.float32).view(1, W, 1)
    c = torch.arange(3, dtype=torch.float32).view(1, 1, 3)
    return y * 1000.0 + x + c * 0.25


def worker(rank, world, port, H, W, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan = StripePlan(H, W, rank, world)
        full = reference_frame(H, W)
        frame = torch.zeros_like(full)
        for s in range(H // 8):
            if s % world == rank:  # what rt_render writes for region {8, rank, world}
                m0 = H - 8 * (s + 1)
                frame[m0:m0 + 8] = full[m0:m0 + 8]
        plan.gather(frame, dist)
        ok = bool(torch.equal(frame, full))
        res = torch.tensor([1 if ok else 0])
        dist.all_reduce(res, op=dist.ReduceOp.MIN)
        if rank == 0:
            with open(out, "w") as f:
                f.write(str(int(res.item())))
    finally:
        dist.destroy_process_group()


def assemble_worker(rank, world, port, H, W, K, out):
    """K frames per sequence: each rank fills its stripes of every frame, `assemble` leaves frame f whole on rank f % world."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan = StripePlan(H, W, rank, world)
        full = torch.stack([reference_frame(H, W) + 0.5 * f for f in range(K)])
        ok = True
        for rep in range(2):  # the second call reuses the plan's buffers
            frames = torch.full_like(full, -1.0)
            for s in range(H // 8):
                if s % world == rank:
                    m0 = H - 8 * (s + 1)
                    frames[:, m0:m0 + 8] = full[:, m0:m0 + 8]
            before = frames.clone()
            roots = plan.assemble(frames, dist)
            ok = ok and roots == [f for f in range(K) if f % world == rank]
            for f in range(K):
                want = full[f] if f % world == rank else before[f]
                ok = ok and bool(torch.equal(frames[f], want))
        res = torch.tensor([1 if ok else 0])
        dist.all_reduce(res, op=dist.ReduceOp.MIN)
        if rank == 0:
            with open(out, "w") as f:
                f.write(str(int(res.item())))
    finally:
        dist.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,H,W", [(2, 64, 40), (2, 72, 33), (3, 40, 16)])
def test_stripe_gather_over_gloo(tmp_path, world, H, W):
    out = tmp_path / "ok.txt"
    mp.spawn(worker, args=(world, free_port(), H, W, str(out)), nprocs=world, join=True)
    assert out.read_text() == "1"


@pytest.mark.parametrize("world,H,W,K", [(2, 64, 40, 4), (2, 72, 33, 3), (3, 40, 16, 5), (3, 40, 16, 2),
                                         (8, 1080, 4, 10), (4, 1080, 4, 5)])  # the headline frame's 135 stripes over 8 and 4 ranks
def test_sequence_of_frames_assembled_on_their_root_ranks(tmp_path, world, H, W, K):
    out = tmp_path / "ok.txt"
    mp.spawn(assemble_worker, args=(world, free_port(), H, W, K, str(out)), nprocs=world, join=True)
    assert out.read_text() == "1"


def test_single_rank_plan_is_identity():
    plan = StripePlan(16, 4, 0, 1)
    f = reference_frame(16, 4)
    g = f.clone()
    plan.gather(g, None)
    assert torch.equal(f, g)
    seq = torch.stack([f, f + 1.0])
    assert plan.assemble(seq, None) == [0, 1] and torch.equal(seq[1], f + 1.0)
    assert plan.owner(0) == 0 and plan.owner(15) == 0


def test_owner_rule_matches_region_semantics():
    plan = StripePlan(64, 8, 1, 4)
    assert [plan.owner(y) for y in (0, 7, 8, 16, 24, 32, 63)] == [0, 0, 1, 2, 3, 0, 3]
    assert plan.mine.tolist() == [1, 5]
    assert plan.my_blocks.tolist() == [6, 2]
    with pytest.raises(ValueError):
        StripePlan(60, 8, 0, 2)
