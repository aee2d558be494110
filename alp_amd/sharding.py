"""Multi-GPU plumbing for the ALP path: static sharding of a column by whole rowgroups + timing reduction.

Vectors are independent given their rowgroup state and rowgroups are fully independent (reference
include/alp/sampler.hpp:16-18 reads only the rowgroup's own <= 100 vectors), so N GPUs = N disjoint contiguous
rowgroup ranges, one process per GPU, NO collective on the data path.  torch.distributed is used
only for the start/stop barrier and the max-over-ranks of the elapsed time (backend "nccl" = RCCL on the GPU box,
"gloo" in the CPU tests)."""
from __future__ import annotations

import time

ROWGROUP_VECTORS = 100


def rowgroup_shard(n_vectors_total: int, rank: int, world: int):
    """(first_vector, n_vectors) of `rank`'s contiguous shard; shards are whole rowgroups, sizes differ by <= 1 rowgroup,
    and every vector belongs to exactly one shard."""
    assert 0 <= rank < world
    n_rg = (n_vectors_total + ROWGROUP_VECTORS - 1) // ROWGROUP_VECTORS
    base, extra = divmod(n_rg, world)
    first_rg = rank * base + min(rank, extra)
    my_rg = base + (1 if rank < extra else 0)
    first_v = first_rg * ROWGROUP_VECTORS
    last_v = min(n_vectors_total, (first_rg + my_rg) * ROWGROUP_VECTORS)
    return first_v, max(0, last_v - first_v)


def timed_steps(step, steps: int, warmup: int, device_sync, dist=None, device=None):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + device sync on both sides; returns the MAX
    over ranks of the elapsed seconds (the driver contract of bench.py)."""
    import torch

    def barrier():
        device_sync()
        if dist is not None:
            dist.barrier()
        device_sync()

    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    return elapsed


def concat_shards(shards):
    """The only "exchange" of the sharded encode: given the per-rank outputs in rank order — tuples
    (rowgroup states, vector descriptors, packed stream, exception stream) as numpy arrays, descriptors with the fields
    `packed_off` / `exc_off` counted from the start of the rank's own streams — returns the same four arrays for the whole
    column: streams concatenated, every rank's offsets shifted by the bytes of the ranks before it.  The result is byte for
    byte what one GPU produces for the unsharded column (tests/test_sharding.py)."""
    import numpy as np
    rgs, vecs, packs, excs = [], [], [], []
    p_off = e_off = 0
    for rg, vec, packed, exc in shards:
        v = vec.copy()
        v["packed_off"] += np.uint64(p_off)
        v["exc_off"] += np.uint64(e_off)
        rgs.append(rg)
        vecs.append(v)
        packs.append(packed)
        excs.append(exc)
        p_off += packed.size
        e_off += exc.size
    return np.concatenate(rgs), np.concatenate(vecs), np.concatenate(packs), np.concatenate(excs)
