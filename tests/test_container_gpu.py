"""GPU: SURVEY §8(f) item 1 — tail padding and the serialized container: a column whose length is not a multiple of
1024 is padded, encoded, serialised to a host blob, deserialised into fresh buffers and decoded back to the original
bits; malformed blobs are rejected before any kernel sees them."""
import numpy as np
import pytest
import torch

import datagen

pytestmark = pytest.mark.gpu


def make(n_values, seed):
    full = np.concatenate([datagen.mixed_column(120, seed=seed), datagen.rd_column(101, seed=seed + 1), datagen.decimal_column(30, 3, seed=seed + 2)])
    return full[:n_values].copy()


@pytest.mark.parametrize("n_values", [251 * 1024, 250 * 1024 + 1, 100 * 1024 + 517, 1023, 1])
def test_tail_padding_and_blob_round_trip(ctx, n_values):
    from alp_amd import capi
    data = make(n_values, seed=n_values % 97)
    n_vectors = (n_values + 1023) // 1024
    x = torch.empty(n_vectors * 1024, dtype=torch.float64, device="cuda")
    x.fill_(float("nan"))  # stale memory past the data must not leak into the encoding
    x[:n_values] = torch.from_numpy(data).cuda()
    ctx.pad_tail(x, n_values)
    ctx.synchronize()
    if n_values % 1024:
        first = (n_values // 1024) * 1024
        assert bool((x[n_values:].view(torch.int64) == x[first].view(torch.int64)).all())
    col = ctx.encode(x)
    blob = ctx.to_blob(col, n_values)
    assert bytes(blob[:7]) == b"ALPGPU1"
    col2, nv = ctx.from_blob(blob)
    assert nv == n_values and col2.n_vectors == n_vectors
    out = ctx.decode(col2)
    ctx.synchronize()
    assert np.array_equal(out[:n_values].cpu().numpy().view(np.uint64), data.view(np.uint64))
    # the blob is a faithful image of the device column
    blob2 = ctx.to_blob(col2, n_values)
    assert np.array_equal(blob, blob2)


def test_malformed_blobs_are_rejected(ctx):
    from alp_amd import capi
    data = make(30 * 1024, seed=5)
    x = torch.from_numpy(data).cuda()
    col = ctx.encode(x)
    blob = ctx.to_blob(col, data.size)
    hdr_words = 8
    n_rg = int(np.frombuffer(blob[:64].tobytes(), np.uint64)[4])
    desc0 = 64 + 32 * n_rg

    def corrupt(fn):
        b = blob.copy()
        fn(b)
        with pytest.raises(capi.AlpGpuError):
            ctx.from_blob(b)

    corrupt(lambda b: b.__setitem__(slice(0, 4), np.frombuffer(b"NOPE", np.uint8)))                 # magic
    corrupt(lambda b: b.view(np.uint64).__setitem__(desc0 // 8 + 4 * 3, np.uint64(1 << 40)))          # packed_off of vector 3 far outside
    corrupt(lambda b: b.__setitem__(desc0 + 32 * 2 + 24, 200))                                        # bw = 200
    corrupt(lambda b: b.view(np.uint64).__setitem__(5, np.uint64(int(b.view(np.uint64)[5]) + 128)))   # packed_bytes > blob
    with pytest.raises(capi.AlpGpuError):
        ctx.from_blob(blob[: blob.size // 2].copy())                                                   # truncated
