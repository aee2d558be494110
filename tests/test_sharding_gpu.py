"""GPU: configs[4] in the small — a column encoded shard by shard (whole-rowgroup shards, as bench.py --gpus N does on N
devices) concatenates to exactly what one encode of the whole column produces, and the bench's on-device generator gives
the same column however it is sharded.  Plus the configs[1] column at FULL size (1 Mi vectors): GPU decode against the
oracle on a sample that covers every bit width, and a checksum over all of it."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_sharded_gpu_encode_concatenates_to_the_unsharded_column(ctx):
    import bench
    from alp_amd import capi, sharding
    bench.MIX_BLOCK_RG = 3  # small generation blocks so that shards cut through them
    try:
        n = 1137
        dev = torch.device("cuda:0")
        whole = bench.mixed_column_shard(0, n, dev, seed=42)
        one = ctx.encode(whole)
        ctx.synchronize()
        for world in (2, 3, 8):
            parts = []
            for rank in range(world):
                first, cnt = sharding.rowgroup_shard(n, rank, world)
                x = bench.mixed_column_shard(first, cnt, dev, seed=42)
                assert torch.equal(x.view(torch.int64), whole[first * 1024:(first + cnt) * 1024].view(torch.int64)), "the generator must not depend on the sharding"
                col = ctx.encode(x)
                ctx.synchronize()
                parts.append(col.to_host())
            got = sharding.concat_shards(parts)
            for a, b, what in zip(got, one.to_host(), ("rowgroup states", "descriptors", "packed stream", "exception stream")):
                assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), (world, what)
    finally:
        bench.MIX_BLOCK_RG = 640


def test_full_size_benchmark_column_decodes_like_the_oracle(ctx, oracle):
    """BASELINE.json configs[1] at its real size: 1 Mi vectors, bit widths 1..53 by rowgroup."""
    import bench
    import layout
    from alp_amd import capi
    n = 1 << 20
    col, vec, _ = bench.build_decode_column(n, 0, seed=42)
    out = ctx.decode(col)
    ctx.synchronize()
    # (1) against the oracle: two vectors of every 53rd .. i.e. rowgroups 0, 97, 194, ... (97 and 53 coprime: all widths)
    rgs = np.arange(0, (n + 99) // 100, 97)
    idx = np.concatenate([rgs * 100, rgs * 100 + 57])
    idx = np.sort(idx[idx < n])
    assert len(set(vec["bw"][idx].tolist())) == 53
    sub = {k: vec[k][idx].copy() for k in ("bw", "e", "f", "base", "exc_cnt", "lbw")}
    sub["scheme"] = vec["scheme"][idx].astype(np.uint8)
    packed = np.zeros((idx.size, 1024), np.uint64)
    p8 = packed.view(np.uint8).reshape(idx.size, 8192)
    for i, v in enumerate(idx):
        o, b = int(vec["packed_off"][v]), int(vec["bw"][v])
        p8[i, : 128 * b] = col.packed[o:o + 128 * b].cpu().numpy()
    sub["packed"] = packed
    sub["packed_left"] = np.zeros((idx.size, 1024), np.uint16)
    sub["exc"] = np.zeros((idx.size, 1024), np.float64)
    sub["pos"] = np.zeros((idx.size, 1024), np.uint16)
    sub["dict"] = np.zeros((idx.size // 100 + 1, 8), np.uint16)
    sub["dict_size"] = np.zeros(idx.size // 100 + 1, np.uint8)
    want = oracle.decode_column(sub)
    got = out.view(-1, 1024)[torch.from_numpy(idx).cuda()].cpu().numpy()
    assert np.array_equal(got.view(np.uint64).reshape(-1), want.view(np.uint64))
    # (2) a checksum over ALL vectors: the decode is deterministic — a second decode (other launch shape) gives the same bits,
    # and every value lies on its vector's decimal grid: (x * 10^(e-f)) is an integer in [base, base + 2^bw)
    ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 2)
    out2 = ctx.decode(col)
    ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)
    ctx.synchronize()
    assert torch.equal(out.view(torch.int64), out2.view(torch.int64))
    s1 = out.view(torch.int64).sum().item()
    s2 = out2.view(torch.int64).sum().item()
    assert s1 == s2
    # every vector's values, scaled back by 10^(e-f) = 100, are integers inside [base, base + 2^bw) wherever that is exact in doubles (bw <= 40)
    narrow = torch.from_numpy((vec["bw"] <= 40)).cuda()
    base = torch.from_numpy(vec["base"].astype(np.float64)).cuda()
    span = torch.from_numpy((2.0 ** vec["bw"].astype(np.float64))).cuda()
    scaled = torch.round(out.view(-1, 1024) * 100.0)
    lo_ok = (scaled >= base[:, None] - 0.5).all(dim=1)
    hi_ok = (scaled < (base + span)[:, None] + 0.5).all(dim=1)
    assert bool((lo_ok & hi_ok)[narrow].all())
