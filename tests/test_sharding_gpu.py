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


@pytest.mark.parametrize("kind", ["mixed", "rd"])
def test_full_size_encode_columns(ctx, oracle, kind):
    """BASELINE.json configs[2] / configs[3] at their real size (1 Mi vectors, bench.py's on-device generator): the size-independent
    properties — the GPU decode of the GPU encode returns the input bits; the single-pass and the two-pass encoders agree on every
    byte of every stream; a second encode reproduces the first — and, on rowgroups spread over the whole column, every descriptor,
    packed word and exception record is the oracle's."""
    import bench
    import layout
    from alp_amd import capi
    n = 1 << 20
    dev = torch.device("cuda:0")
    x = bench.synthetic_input(kind, n, dev, seed=42)

    def encode():
        col = capi.DeviceColumn(n, 0, packed_capacity=int(n * 8192 * 0.95) + 4096, exc_capacity=int(n * 8192 * 0.25) + 4096)
        ctx.encode(x, col)
        ctx.synchronize()
        pb, eb, ov = ctx.column_totals(col)
        assert ov == 0
        return col, pb, eb

    col, pb, eb = encode()
    out = ctx.decode(col)
    ctx.synchronize()
    assert torch.equal(out.view(torch.int64), x.view(torch.int64)), "round trip"
    del out

    def same(a, b, pb, eb):
        return (torch.equal(a.vectors, b.vectors) and torch.equal(a.rowgroups, b.rowgroups) and torch.equal(a.packed[:pb], b.packed[:pb])
                and torch.equal(a.exc[:eb], b.exc[:eb]))

    again, pb2, eb2 = encode()
    assert (pb2, eb2) == (pb, eb) and same(col, again, pb, eb), "run to run"
    del again
    try:
        ctx.set_option(capi.OPT_ENCODE_TWO_PASS, 1)
        two, pb3, eb3 = encode()
    finally:
        ctx.set_option(capi.OPT_ENCODE_TWO_PASS, 0)
    assert (pb3, eb3) == (pb, eb) and same(col, two, pb, eb), "single pass vs two pass"
    del two
    # the oracle on every 953rd rowgroup (a rowgroup's encoding depends on nothing outside it)
    vec_all = col.vectors.cpu().numpy().view(capi.VECTOR_DTYPE)[:n]
    rg_all = col.rowgroups.cpu().numpy().view(capi.ROWGROUP_DTYPE)[: (n + 99) // 100]
    for g in range(0, n // 100, 953):
        v0, v1 = g * 100, g * 100 + 100
        vec = vec_all[v0:v1].copy()
        p0, e0 = int(vec["packed_off"][0]), int(vec["exc_off"][0])
        p1 = int(vec_all["packed_off"][v1]) if v1 < n else pb
        e1 = int(vec_all["exc_off"][v1]) if v1 < n else eb
        packed = col.packed[p0:p1].cpu().numpy()
        exc = col.exc[e0:e1].cpu().numpy()
        vec["packed_off"] -= p0
        vec["exc_off"] -= e0
        got = layout.expand(rg_all[g:g + 1], vec, packed, exc)
        want = oracle.encode_column(x[v0 * 1024:v1 * 1024].cpu().numpy())
        for k in ("scheme", "e", "f", "bw", "lbw", "base", "exc_cnt"):
            assert np.array_equal(got[k], want[k]), (kind, g, k)
        assert np.array_equal(got["packed"], want["packed"]) and np.array_equal(got["packed_left"], want["packed_left"]), (kind, g, "packed words")
        for v in range(100):
            c = int(want["exc_cnt"][v])
            assert np.array_equal(got["pos"][v, :c], want["pos"][v, :c]), (kind, g, v, "positions")
            w = np.uint64 if want["scheme"][v] == 2 else np.uint16
            assert np.array_equal(got["exc"][v].view(w)[:c], want["exc"][v].view(w)[:c]), (kind, g, v, "exception values")


def test_one_context_per_visible_device_in_one_process():
    """The in-process multi-GPU path (alpgpu_compress_host_multi_* / alpgpu_decompress_host_multi_*) on EVERY GPU the box has: one context per
    device, one host column cut into whole-rowgroup shards, ONE blob — byte for byte the blob of device 0 alone — and a per-device encode /
    decode round trip issued from this one thread in turn (every entry point makes its context's device current itself:
    tools/audit_set_device.py).  Needs two GPUs: skipped on the one-GPU test boxes, armed for the first node that has more."""
    import datagen
    from alp_amd import capi
    n_dev = torch.cuda.device_count()
    if n_dev < 2:
        pytest.skip(f"{n_dev} GPU visible: the multi-device form needs two")
    ctxs = [capi.Context(d) for d in range(n_dev)]  # each follows torch's current stream of ITS device
    for d, c in enumerate(ctxs):
        assert "gfx950" in c.device_info()["name"], (d, c.device_info())
    n = 100 * 3 * n_dev + 57
    a = datagen.mixed_column(n, seed=77, exc_rate=0.01)
    a[: 200 * 1024] = np.random.default_rng(1).random(200 * 1024)  # two ALP_RD rowgroups
    x = torch.from_numpy(a)
    want = ctxs[0].compress_host(x)
    blob = capi.Context.compress_host_multi(ctxs, x)
    assert torch.equal(blob, want), "the shards of all devices join into the one-device blob"
    out = torch.empty(n * 1024, dtype=torch.float64)
    assert capi.Context.decompress_host_multi(ctxs, blob, out) == n * 1024
    assert torch.equal(out.view(torch.int64), x.view(torch.int64))
    # device-resident columns on every device, driven from this thread in turn (the current device changes under each call)
    cols = []
    for d, c in enumerate(ctxs):
        with torch.cuda.device(d):
            xd = x.to(f"cuda:{d}")
        cols.append((xd, c.encode(xd)))
    for d, c in reversed(list(enumerate(ctxs))):
        xd, col = cols[d]
        back = c.decode(col)
        c.synchronize()
        assert back.device.index == d and torch.equal(back.view(torch.int64), xd.view(torch.int64)), f"device {d}"
        host = col.to_host()
        for got, ref, what in zip(host, cols[0][1].to_host(), ("rowgroup states", "descriptors", "packed stream", "exception stream")):
            assert np.array_equal(got.view(np.uint8), ref.view(np.uint8)), (d, what)
