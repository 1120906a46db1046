"""CPU, world_size 2 over gloo: region sharding + the gather of predictions (the N>1 path of bench.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pepper_thesis_amd.dist import GatherCapacityError, gather_predictions, shard_regions


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_items = 7
        mine = shard_regions(n_items, rank, world)
        # ragged: region i yields i+1 windows; row value encodes (region, window)
        rows, keys = [], []
        for i in mine:
            for k in range(i + 1):
                rows.append([float(i), float(k), float(i * 100 + k)])
                keys.append(i * 1000 + k)
        local = torch.tensor(rows, dtype=torch.float32).reshape(-1, 3)
        res = gather_predictions(local, dst=0, keys=torch.tensor(keys, dtype=torch.int64))
        if rank == 0:
            allrows, allkeys, counts = res
            q.put((allrows.numpy().tolist(), allkeys.numpy().tolist(), counts))
        else:
            assert res is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_shard_is_a_partition():
    for world in (1, 2, 3, 8):
        seen = sorted(i for r in range(world) for i in shard_regions(23, r, world))
        assert seen == list(range(23))


@pytest.mark.timeout(120)
def test_gather_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    rows, keys, counts = q.get(timeout=100)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # every (region, window) pair exactly once; rank-major order; counts per rank
    assert counts == [sum(i + 1 for i in range(7) if i % 2 == r) for r in range(2)]
    assert sorted(keys) == sorted(i * 1000 + k for i in range(7) for k in range(i + 1))
    for row, key in zip(rows, keys):
        assert row[2] == (key // 1000) * 100 + key % 1000


def test_gather_single_process_is_identity():
    x = torch.arange(12, dtype=torch.float32).reshape(4, 3)
    rows, keys, counts = gather_predictions(x)
    assert rows is x and counts == [4]


def _worker_capacity(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # ragged: rank 0 holds 2 rows, rank 1 holds 9; the destination has room for 2 * world = 4 (the old default bound)
        local = torch.full((2 if rank == 0 else 9, 3), float(rank))
        try:
            gather_predictions(local, dst=0, capacity_rows=4 if rank == 0 else None)
            q.put((rank, "no error"))
        except GatherCapacityError as e:
            q.put((rank, (e.total, e.capacity, e.counts)))
        # both ranks are still in step: the next collective completes, and an adequate capacity succeeds
        res = gather_predictions(local, dst=0, capacity_rows=11 if rank == 0 else None)
        if rank == 0:
            q.put(("ok", res[2]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gather_capacity_error_is_collective_world2_gloo():
    """pv_gather's rule (csrc/pv_comm.hip) on its torch.distributed twin: over capacity fails on EVERY rank, nobody hangs"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_capacity, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=100) for _ in range(3)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    errs = {k: v for k, v in got if k in (0, 1)}
    assert errs == {0: (11, 4, [2, 9]), 1: (11, 4, [2, 9])}
    assert ("ok", [2, 9]) in got
