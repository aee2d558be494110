"""GPU: alpgpu_compress_host_* / alpgpu_decompress_host_* — a column and its serialized form in HOST memory, through the two-stream
chunked pipeline.  The blob must be byte for byte what a one-piece encode of the same values on the device serializes to
(alpgpu_encode_* + alpgpu_column_to_blob), whatever the number of chunks and however the column ends; decompression returns the
input bits; blobs travel both ways between the two routes; corrupt blobs and short buffers are refused."""
import numpy as np
import pytest
import torch

import datagen
from alp_amd import capi

pytestmark = pytest.mark.gpu
CHUNK = 12800  # kHostChunkVectors


def one_piece_blob(ctx, x_host):
    """encode on the device in one piece, serialize: the reference the pipeline must reproduce"""
    n_values = x_host.numel()
    n = (n_values + 1023) // 1024
    dt = x_host.dtype
    xd = torch.zeros(n * 1024, dtype=dt, device="cuda")
    xd[:n_values] = x_host.cuda()
    ctx.pad_tail(xd, n_values)
    col = ctx.encode(xd)
    ctx.synchronize()
    return ctx.to_blob(col, n_values)


def column(n_values, dtype, seed):
    n = (n_values + 1023) // 1024
    if dtype == torch.float64:
        a = datagen.mixed_column(n, seed=seed, exc_rate=0.01)
        k = min(300, n - n // 3)
        a[(n // 3) * 1024:(n // 3 + k) * 1024] = np.random.default_rng(seed).random(k * 1024)  # some ALP_RD rowgroups
        return torch.from_numpy(a[:n_values].copy())
    rng = np.random.default_rng(seed)
    a = np.round(rng.uniform(-1e3, 1e3, n * 1024), 2).astype(np.float32)
    k = min(300, n - n // 3)
    a[(n // 3) * 1024:(n // 3 + k) * 1024] = rng.random(k * 1024).astype(np.float32)
    return torch.from_numpy(a[:n_values].copy())


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("n_values", [1, 1024, 100 * 1024 + 17, CHUNK * 1024, 2 * CHUNK * 1024 + 5 * 1024 + 333, 3 * CHUNK * 1024 - 1])
def test_pipeline_blob_equals_the_one_piece_blob_and_round_trips(dtype, n_values):
    ctx = capi.Context(0)
    x = column(n_values, dtype, seed=n_values % 1000 + 7)
    xp = torch.empty(n_values, dtype=dtype, pin_memory=True)
    xp.copy_(x)
    blob = ctx.compress_host(xp)
    want = one_piece_blob(ctx, x)
    got = blob.numpy()
    want = np.asarray(want).view(np.uint8).reshape(-1)
    assert got.size == want.size
    assert np.array_equal(got, want), f"first difference at byte {int(np.nonzero(got != want)[0][0])}"
    out = torch.full((n_values + 8,), 7, dtype=dtype, pin_memory=True)
    assert ctx.decompress_host(blob, out) == n_values
    it = torch.int64 if dtype == torch.float64 else torch.int32
    assert torch.equal(out[:n_values].view(it), x.view(it))
    assert bool((out[n_values:] == 7).all()), "nothing is written past the column's values"
    # pageable memory on both sides gives the same bytes
    blob2 = ctx.compress_host(x.clone())
    assert torch.equal(blob2, blob)


def test_empty_column_and_refusals():
    ctx = capi.Context(0)
    empty = torch.empty(0, dtype=torch.float64)
    blob = ctx.compress_host(empty)
    assert blob.numel() == 64
    assert ctx.decompress_host(blob, torch.empty(0, dtype=torch.float64)) == 0
    x = torch.from_numpy(datagen.decimal_column(230, 2, seed=5))
    blob = ctx.compress_host(x)
    with pytest.raises(capi.AlpGpuError):  # output too small
        ctx.decompress_host(blob, torch.empty(x.numel() - 1, dtype=torch.float64))
    with pytest.raises(capi.AlpGpuError):  # the other value type
        ctx.decompress_host(blob, torch.empty(x.numel(), dtype=torch.float32))
    with pytest.raises(capi.AlpGpuError):  # blob buffer too small
        ctx.compress_host(x, torch.empty(1000, dtype=torch.uint8))
    bad = blob.clone()
    bad[64 + 32 * 3 + 32 * 5 + 8] ^= 0x7F  # an exception offset of vector 5
    with pytest.raises(capi.AlpGpuError):
        ctx.decompress_host(bad, torch.empty(x.numel(), dtype=torch.float64))
    with pytest.raises(capi.AlpGpuError):  # truncated
        ctx.decompress_host(blob[: blob.numel() - 9], torch.empty(x.numel(), dtype=torch.float64))
    # a stream after the pipeline's own: the context's stream is what it was
    y = torch.from_numpy(datagen.decimal_column(10, 1, seed=6)).cuda()
    col = ctx.encode(y)
    back = ctx.decode(col)
    ctx.synchronize()
    assert torch.equal(back.view(torch.int64), y.view(torch.int64))


def test_a_record_outside_its_chunks_stream_range_is_refused():
    """the chunked route uploads only the byte range between a chunk's first offsets and the next chunk's; a descriptor
    that points into ANOTHER chunk's range (inside the whole stream, so alpgpu_column_from_blob decodes it deterministically) must not be
    decoded from bytes that have not been uploaded — the chunked route refuses the blob."""
    ctx = capi.Context(0)
    n = CHUNK + 300  # two chunks
    x = torch.from_numpy(datagen.decimal_column(n, 2, seed=17))
    blob = ctx.compress_host(x)
    out = torch.empty(x.numel(), dtype=torch.float64)
    assert ctx.decompress_host(blob, out) == x.numel()
    nrg = (n + 99) // 100
    vec = blob.numpy()[64 + 32 * nrg: 64 + 32 * nrg + 32 * n].view(capi.VECTOR_DTYPE)
    twin = int(np.nonzero(vec["bw"][:CHUNK] == vec[CHUNK + 50]["bw"])[0][0])  # same record size: the whole-stream check cannot see the swap
    bad = blob.clone()
    bvec = bad.numpy()[64 + 32 * nrg: 64 + 32 * nrg + 32 * n].view(capi.VECTOR_DTYPE)
    bvec[CHUNK + 50]["packed_off"] = vec[twin]["packed_off"]  # a vector of chunk 1 reads chunk 0's bytes
    col, n_values = ctx.from_blob(bad.numpy())  # the one-piece route holds the whole stream: accepted, deterministic
    assert n_values == x.numel()
    with pytest.raises(capi.AlpGpuError, match="chunk"):
        ctx.decompress_host(bad, out)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_several_contexts_write_the_one_context_blob(dtype):
    """alpgpu_compress_host_multi_* / alpgpu_decompress_host_multi_* through the C ABI: 1, 2, 3 and 5 contexts (all on GPU 0 here), pinned and pageable"""
    ctxs = [capi.Context(0) for _ in range(5)]
    n_values = 2 * CHUNK * 1024 + 7 * 1024 + 99
    x = column(n_values, dtype, seed=31)
    want = ctxs[0].compress_host(x)
    for k in (1, 2, 3, 5):
        blob = capi.Context.compress_host_multi(ctxs[:k], x)
        assert torch.equal(blob, want), f"{k} contexts"
        out = torch.full((n_values + 3,), 9, dtype=dtype, pin_memory=(k % 2 == 0))
        assert capi.Context.decompress_host_multi(ctxs[:k], blob, out) == n_values
        it = torch.int64 if dtype == torch.float64 else torch.int32
        assert torch.equal(out[:n_values].view(it), x.view(it)) and bool((out[n_values:] == 9).all())
    # a corrupt descriptor in the LAST shard is found by the context that owns it
    nrg = ((n_values + 1023) // 1024 + 99) // 100
    bad = want.clone()
    n = (n_values + 1023) // 1024
    bad.numpy()[64 + 32 * nrg: 64 + 32 * nrg + 32 * n].view(capi.VECTOR_DTYPE)[n - 5]["bw"] = 77
    with pytest.raises(capi.AlpGpuError):
        capi.Context.decompress_host_multi(ctxs[:3], bad, torch.empty(n_values, dtype=dtype))


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_more_contexts_than_rowgroups_and_a_short_buffer(dtype):
    """with more contexts than rowgroups the trailing shards are empty; their regions must not reach past the caller's buffer.
    An incompressible column of 150 vectors (2 rowgroups, n % 100 != 0) over 5 contexts, into a buffer that is too small: the call fails with
    ALPGPU_ERR_CAPACITY, writes nothing behind the buffer (a guard band stays intact), and the size it reports is one that works."""
    import ctypes as C
    ctxs = [capi.Context(0) for _ in range(5)]
    n_values = 150 * 1024 - 7
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.random(n_values).astype(np.float64 if dtype == torch.float64 else np.float32))  # ALP_RD: ~7/8 of the input stays
    t = "f64" if dtype == torch.float64 else "f32"
    want = ctxs[0].compress_host(x)
    small = int(want.numel() * 0.6) // 8 * 8
    guard = 1 << 16
    buf = torch.full((small + guard,), 0xAB, dtype=torch.uint8)
    arr = (C.c_void_p * 5)(*[c.h for c in ctxs])
    w = C.c_uint64()
    rc = getattr(capi.lib, "alpgpu_compress_host_multi_" + t)(arr, 5, C.c_void_p(x.data_ptr()), n_values, C.c_void_p(buf.data_ptr()), small, C.byref(w))
    assert rc != 0 and b"small" in capi.lib.alpgpu_last_error()
    assert bool((buf[small:] == 0xAB).all()), "bytes behind the caller's capacity were written"
    assert w.value > small
    big = torch.empty(w.value, dtype=torch.uint8)
    blob = capi.Context.compress_host_multi(ctxs, x, big)  # the reported size suffices
    assert torch.equal(blob, want)
    out = torch.empty(n_values, dtype=dtype)
    assert capi.Context.decompress_host_multi(ctxs, blob, out) == n_values
    it = torch.int64 if dtype == torch.float64 else torch.int32
    assert torch.equal(out.view(it), x.view(it))
