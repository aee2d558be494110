"""GPU: alpgpu_encode_* with the rowgroup search BESIDE the vector encode (ALPGPU_OPT_ENCODE_ASYNC_INIT, the default for long double columns):
a persistent search kernel on the context's second stream publishes the rowgroup states while the single-pass encode polls for them.
Everything the call writes — rowgroup states (tags cleared), descriptors, both streams, totals, the ALP_RD order tables' effect on the
packed bytes — must be byte for byte what the search-in-front route writes, also when the single pass is forced into its recovery route,
and must agree with the oracle where the oracle is run."""
import numpy as np
import pytest
import torch

import datagen
import layout
from alp_amd import capi

pytestmark = pytest.mark.gpu
N_VECTORS = 104_857  # 1049 rowgroups (the async route starts at 1024), the last one partial


def long_column(dtype):
    """mixed decimals with ALP_RD stretches, built on the device from a few host-made pieces"""
    if dtype == "f64":
        pieces = [datagen.mixed_column(1000, seed=s, exc_rate=0.02) for s in (1, 2)] + [datagen.rd_column(300, seed=3), datagen.drifting_column(700, seed=4)]
    else:
        pieces = [datagen.mixed_column_f32(1000, seed=s, exc_rate=0.02) for s in (1, 2)] + [datagen.rd_column_f32(300, seed=3), datagen.drifting_column_f32(700, seed=4)]
    host = np.concatenate(pieces)  # 3000 vectors
    dev = torch.from_numpy(host).cuda()
    reps = (N_VECTORS * 1024 + dev.numel() - 1) // dev.numel()
    x = dev.repeat(reps)[: N_VECTORS * 1024].contiguous()
    # make the repeats differ: a rowgroup-dependent offset that keeps the decimals
    x.view(-1, 1024)[:, 0] += torch.arange(N_VECTORS, device="cuda", dtype=x.dtype) % 7
    return x, host


def encode_with(ctx, x, dtype, mode, force_stall=False):
    ctx.set_option(capi.OPT_ENCODE_ASYNC_INIT, mode)
    ctx.set_option(capi.OPT_DEBUG_FORCE_STALL, 1 if force_stall else 0)
    try:
        col = capi.DeviceColumn(N_VECTORS, 0, dtype=dtype)
        col.rowgroups.fill_(0x5A)  # dirty buffers: stale tags, stale descriptors
        col.vectors.fill_(0x5A)
        ctx.encode(x, col)
        ctx.synchronize()
        pb, eb, ov = ctx.column_totals(col)
        assert ov == 0
        return col, pb, eb
    finally:
        ctx.set_option(capi.OPT_ENCODE_ASYNC_INIT, 1)
        ctx.set_option(capi.OPT_DEBUG_FORCE_STALL, 0)


def same_column(a, b):
    (ca, pa, ea), (cb, pb_, eb_) = a, b
    assert (pa, ea) == (pb_, eb_)
    assert torch.equal(ca.rowgroups, cb.rowgroups), "rowgroup states differ"
    assert torch.equal(ca.vectors, cb.vectors), "descriptors differ"
    assert torch.equal(ca.packed[:pa], cb.packed[:pa]) and torch.equal(ca.exc[:ea], cb.exc[:ea]), "streams differ"


@pytest.mark.parametrize("dtype,async_mode", [("f64", 1), ("f32", 2)])
def test_search_beside_the_encode_writes_the_same_column(ctx, dtype, async_mode):
    x, _ = long_column(dtype)
    front = encode_with(ctx, x, dtype, 0)
    for rep in range(3):  # the hand-over between two kernels is a race if it is wrong: more than one try
        beside = encode_with(ctx, x, dtype, async_mode)
        same_column(front, beside)
    rg = beside[0].rowgroups.cpu().numpy().view(capi.ROWGROUP_DTYPE)[: (N_VECTORS + 99) // 100]
    assert not rg["pad"].any(), "the publishing tags are cleared: the states are the reference's bytes"
    assert (rg["scheme"] == capi.SCHEME_ALP).any() and (rg["scheme"] != capi.SCHEME_ALP).any()
    out = ctx.decode(beside[0])
    it = torch.int64 if dtype == "f64" else torch.int32
    assert torch.equal(out.view(it), x.view(it))
    # and the recovery route behind a search that ran beside the (stalled) single pass
    same_column(front, encode_with(ctx, x, dtype, async_mode, force_stall=True))


def test_search_beside_the_encode_of_a_mostly_alp_rd_column(ctx):
    """a column whose head is mostly ALP_RD: the persistent search keeps all three of its workgroups per CU (k_rowgroup_init: `walkers`) and walks
    the rowgroups with the larger stride; same column as with the search in front, three times over, and through the recovery route"""
    pieces = [datagen.rd_column(1200, seed=11), datagen.mixed_column(300, seed=12, exc_rate=0.02), datagen.rd_column(1300, seed=13, kind="latlon"), datagen.drifting_column(200, seed=14)]
    dev = torch.from_numpy(np.concatenate(pieces)).cuda()
    reps = (N_VECTORS * 1024 + dev.numel() - 1) // dev.numel()
    x = dev.repeat(reps)[: N_VECTORS * 1024].contiguous()
    front = encode_with(ctx, x, "f64", 0)
    rg = front[0].rowgroups.cpu().numpy().view(capi.ROWGROUP_DTYPE)[: (N_VECTORS + 99) // 100]
    assert (rg["scheme"][:256] != capi.SCHEME_ALP).mean() > 0.5 and (rg["scheme"] == capi.SCHEME_ALP).any()
    for rep in range(3):
        same_column(front, encode_with(ctx, x, "f64", 1))
    same_column(front, encode_with(ctx, x, "f64", 1, force_stall=True))
    out = ctx.decode(front[0])
    assert torch.equal(out.view(torch.int64), x.view(torch.int64))


def test_search_beside_the_encode_agrees_with_the_oracle(ctx, oracle):
    x, _ = long_column("f64")
    col, pb, eb = encode_with(ctx, x, "f64", 1)
    rgs, vec, packed, exc = col.to_host()
    for rg0 in (0, 511, 1047):  # whole rowgroups spread over the column (1048 is the partial last one)
        v0, v1 = rg0 * 100, rg0 * 100 + 100
        want = oracle.encode_column(x[v0 * 1024:v1 * 1024].cpu().numpy())
        got = vec[v0:v1]
        for k in ("scheme", "e", "f", "bw", "lbw", "base", "exc_cnt"):
            assert np.array_equal(got[k].astype(np.int64), want[k].astype(np.int64)), (rg0, k)
        # the rowgroup's packed bytes: the oracle's streams, piece by piece
        w_rg, w_vec, w_packed, w_exc = layout.compact(want)
        p0 = int(got["packed_off"][0])
        assert np.array_equal(packed[p0:p0 + w_packed.size], w_packed), rg0
        e0 = int(got["exc_off"][0])
        assert np.array_equal(exc[e0:e0 + w_exc.size], w_exc), rg0


def test_short_columns_and_the_option(ctx):
    """below 1024 rowgroups nothing changes; the option takes 0 / 1 / 2 only"""
    x = torch.from_numpy(datagen.mixed_column(300, seed=9)).cuda()
    a = ctx.encode(x)
    ctx.set_option(capi.OPT_ENCODE_ASYNC_INIT, 0)
    b = ctx.encode(x)
    ctx.set_option(capi.OPT_ENCODE_ASYNC_INIT, 1)
    ctx.synchronize()
    assert torch.equal(a.vectors, b.vectors) and torch.equal(a.rowgroups, b.rowgroups)
    with pytest.raises(capi.AlpGpuError):
        ctx.set_option(capi.OPT_ENCODE_ASYNC_INIT, 3)
