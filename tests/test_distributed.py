"""N > 1 path on the CPU: the pair sharding and the rank aggregation bench.py uses, with 2 gloo ranks."""
import os
import socket

import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, out_q):
    import torch.distributed as dist

    from dvo_slam_amd import sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    mine = sharding.shard_indices(n_items, rank, world)
    weak = sharding.shard_weak(5, rank, world)
    dist.barrier()
    elapsed = 0.25 + 0.5 * rank  # rank 1 is the straggler
    t_max, n_total = sharding.aggregate(elapsed, len(mine), dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, (mine, weak))
    out_q.put((rank, t_max, n_total, gathered))
    dist.destroy_process_group()


def test_shard_functions():
    from dvo_slam_amd import sharding

    for n in (0, 1, 7, 64):
        for world in (1, 2, 3, 8):
            parts = [sharding.shard_indices(n, r, world) for r in range(world)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert sharding.shard_weak(4, 2, 4) == [8, 9, 10, 11]
    assert [len(p) for p in sharding.split_for_threads(list(range(10)), 4)] == [3, 3, 2, 2]
    assert sharding.split_for_threads([1], 4) == [[1]]
    with pytest.raises(ValueError):
        sharding.shard_indices(4, 2, 2)
    assert sharding.aggregate(1.5, 7, None) == (1.5, 7)


def test_two_gloo_ranks_partition_and_aggregate():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world, n_items = 2, 33
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, t_max, n_total, gathered in results:
        assert t_max == 0.75  # MAX over ranks, as the bench contract asks
        assert n_total == n_items
        all_idx = sorted(i for mine, _ in gathered for i in mine)
        assert all_idx == list(range(n_items))  # every pair aligned exactly once
        weak = [w for _, w in gathered]
        assert weak == [[0, 1, 2, 3, 4], [5, 6, 7, 8, 9]]
