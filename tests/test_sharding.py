"""CPU: the N > 1 path — rowgroup sharding covers a column exactly once, the world_size-2 timing harness (gloo here,
RCCL on the GPU box) returns the max over ranks on every rank, and the per-rank encodes of a sharded column concatenate
(alp_amd.sharding.concat_shards) to exactly the unsharded column's records and streams (oracle on both sides here; the GPU
side of the same statement is tests/test_sharding_gpu.py)."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alp_amd import sharding  # noqa: E402


@pytest.mark.parametrize("n_vectors,world", [(1, 1), (99, 2), (100, 2), (101, 2), (1 << 20, 8), (12_207_031, 8), (250, 4), (7, 8)])
def test_shards_partition_the_column(n_vectors, world):
    nxt = 0
    sizes = []
    for r in range(world):
        first, n = sharding.rowgroup_shard(n_vectors, r, world)
        if n:
            assert first == nxt and first % 100 == 0
            nxt = first + n
        sizes.append(n)
    assert nxt == n_vectors and sum(sizes) == n_vectors
    full = [s for s in sizes if s]
    assert max(full) - min(full) < 200 or len(full) < world


def _worker(rank, world, port, q):
    import time
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, n = sharding.rowgroup_shard(1000, rank, world)
    done = torch.zeros(1000, dtype=torch.int32)

    def step():
        done[first:first + n] += 1
        time.sleep(0.01 * (rank + 1))  # rank 1 is slower: the reported time must be ITS time on both ranks

    el = sharding.timed_steps(step, steps=3, warmup=1, device_sync=lambda: None, dist=dist)
    dist.all_reduce(done)
    q.put((rank, el, int(done.min()), int(done.max())))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_timing_and_coverage():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, e0, mn0, mx0), (r1, e1, mn1, mx1) = res
    assert abs(e0 - e1) < 1e-9, "every rank must report the max over ranks"
    assert e0 >= 3 * 0.02 * 0.9
    assert mn0 == mx0 == 4, "each vector processed exactly once per step (warm-up + 3 steps) by exactly one rank"


def _encode_worker(rank, world, port, q):
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import datagen
    import layout
    from oracle.pyoracle import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    column = np.concatenate([datagen.mixed_column(230, seed=5, exc_rate=0.02), datagen.rd_column(100, seed=6), datagen.mixed_column(137, seed=7)])  # 467 vectors
    n = column.size // 1024
    first, cnt = sharding.rowgroup_shard(n, rank, world)
    mine = layout.compact(Oracle().encode_column(column[first * 1024:(first + cnt) * 1024]))
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)  # the host-side concat of (offset, length) pairs: no device collective
    if rank == 0:
        whole = layout.compact(Oracle().encode_column(column))
        got = sharding.concat_shards(gathered)
        q.put(all(np.array_equal(a.view(np.uint8), b.view(np.uint8)) for a, b in zip(got, whole)))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_sharded_encode_concatenates_to_the_unsharded_column():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    ps = [ctx.Process(target=_encode_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    ok = q.get(timeout=300)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok, "per-rank streams, concatenated with shifted offsets, must equal the single-rank streams byte for byte"
