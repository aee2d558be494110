"""GPU: decode fused into a SUM consumer.  The kernel's summation order is documented in
include/alpgpu.h; reproducing that order on the host from the (bit-exact) decoded values must give the same bits."""
import numpy as np
import pytest
import torch

import datagen
import layout

pytestmark = pytest.mark.gpu


def host_sums(values):
    """values [n, 1024] -> the documented order: lane partials, butterfly over lanes, (w0 + w1) + (w2 + w3)"""
    n = values.shape[0]
    v = values.reshape(n, 4, 2, 64, 2)  # vector, wavefront q, step mm, lane L, pair element
    p = np.zeros((n, 4, 64))
    with np.errstate(invalid="ignore", over="ignore"):
        for mm in range(2):
            p = p + v[:, :, mm, :, 0]
            p = p + v[:, :, mm, :, 1]
        idx = np.arange(64)
        for d in (32, 16, 8, 4, 2, 1):
            p = p + p[:, :, idx ^ d]
        w = p[:, :, 0]
        return (w[:, 0] + w[:, 1]) + (w[:, 2] + w[:, 3])


COLUMNS = {
    "decimal2": lambda: datagen.decimal_column(130, 2, seed=1),
    "mixed_specials": lambda: datagen.mixed_column(120, seed=3, exc_rate=0.02),  # NaN / Inf exceptions propagate into the sums
    "rd": lambda: datagen.rd_column(101, seed=5),
    "odd_count": lambda: datagen.decimal_column(7, 1, seed=9),
}


@pytest.mark.parametrize("vectors_per_wg", [1, 2])
@pytest.mark.parametrize("name", list(COLUMNS.keys()))
def test_decode_sum_matches_documented_order(ctx, oracle, name, vectors_per_wg):
    from alp_amd import capi
    col = COLUMNS[name]()
    enc = oracle.encode_column(col)
    dcol = capi.DeviceColumn.from_host(*layout.compact(enc))
    try:
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vectors_per_wg)
        got = ctx.decode_sum(dcol)
        dec = ctx.decode(dcol)
        ctx.synchronize()
    finally:
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)
    dec = dec.cpu().numpy()
    assert np.array_equal(dec.view(np.uint64), col.view(np.uint64))
    want = host_sums(dec.reshape(-1, 1024))
    got = got.cpu().numpy()
    same = (got.view(np.uint64) == want.view(np.uint64)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), f"{name}: {np.nonzero(~same)[0][:5]} {got[~same][:3]} {want[~same][:3]}"
    finite = np.isfinite(want)
    assert np.allclose(got[finite], col.reshape(-1, 1024)[finite].sum(axis=1), rtol=1e-12, atol=0)


@pytest.mark.parametrize("name", list(COLUMNS.keys()))
def test_decode_count_range_matches_numpy(ctx, oracle, name):
    """the predicate consumer: per-vector counts of lo <= x <= hi on the decoded values (exceptions, NaN, Inf, -0.0 included)"""
    from alp_amd import capi
    col = COLUMNS[name]()
    enc = oracle.encode_column(col)
    dcol = capi.DeviceColumn.from_host(*layout.compact(enc))
    v = col.reshape(-1, 1024)
    finite = col[np.isfinite(col)]
    for lo, hi in ((float(np.quantile(finite, 0.25)), float(np.quantile(finite, 0.75))), (0.0, 0.0), (-np.inf, np.inf), (1.0, -1.0),
                   (float(finite.max()), float(finite.max()))):
        got = ctx.decode_count_range(dcol, lo, hi)
        ctx.synchronize()
        with np.errstate(invalid="ignore"):
            want = ((v >= lo) & (v <= hi)).sum(axis=1)
        assert np.array_equal(got.cpu().numpy().astype(np.int64), want), (name, lo, hi)


# ---- float columns (alpgpu_decode_sum_f32 / alpgpu_decode_count_range_f32) -------------------------------------------------

def host_sums_f32(values):
    """values [n, 1024] float32 -> the order documented in include/alpgpu.h: thread t adds values 4t..4t+3 (in double),
    butterfly over the 64 threads of a wavefront, (w0 + w1) + (w2 + w3)"""
    n = values.shape[0]
    v = values.astype(np.float64).reshape(n, 4, 64, 4)  # vector, wavefront, lane, quad element
    p = np.zeros((n, 4, 64))
    with np.errstate(invalid="ignore", over="ignore"):
        for c in range(4):
            p = p + v[:, :, :, c]
        idx = np.arange(64)
        for d in (32, 16, 8, 4, 2, 1):
            p = p + p[:, :, idx ^ d]
        w = p[:, :, 0]
        return (w[:, 0] + w[:, 1]) + (w[:, 2] + w[:, 3])


COLUMNS_F32 = {
    "decimal1": lambda: datagen.decimal_column_f32(130, 1, seed=1),
    "mixed_specials": lambda: datagen.mixed_column_f32(120, seed=3, exc_rate=0.02),
    "rd": lambda: datagen.rd_column_f32(101, seed=5),
    "odd_count": lambda: datagen.decimal_column_f32(7, 2, seed=9),
}


@pytest.fixture(scope="module")
def of32():
    from oracle.pyoracle import OracleF32
    return OracleF32()


@pytest.mark.parametrize("name", list(COLUMNS_F32.keys()))
def test_decode_sum_f32_matches_documented_order(ctx, of32, name):
    from alp_amd import capi
    col = COLUMNS_F32[name]()
    enc = of32.encode_column(col)
    dcol = capi.DeviceColumn.from_host(*layout.compact(enc, 4), dtype="f32")
    got = ctx.decode_sum(dcol)
    dec = ctx.decode(dcol)
    ctx.synchronize()
    dec = dec.cpu().numpy()
    assert np.array_equal(dec.view(np.uint32), col.view(np.uint32))
    want = host_sums_f32(dec.reshape(-1, 1024))
    got = got.cpu().numpy()
    same = (got.view(np.uint64) == want.view(np.uint64)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), f"{name}: {np.nonzero(~same)[0][:5]} {got[~same][:3]} {want[~same][:3]}"


@pytest.mark.parametrize("name", list(COLUMNS_F32.keys()))
def test_decode_count_range_f32_matches_numpy(ctx, of32, name):
    from alp_amd import capi
    col = COLUMNS_F32[name]()
    enc = of32.encode_column(col)
    dcol = capi.DeviceColumn.from_host(*layout.compact(enc, 4), dtype="f32")
    v = col.reshape(-1, 1024)
    finite = col[np.isfinite(col)]
    for lo, hi in ((float(np.quantile(finite, 0.25)), float(np.quantile(finite, 0.75))), (0.0, 0.0), (-np.inf, np.inf), (1.0, -1.0),
                   (float(finite.max()), float(finite.max()))):
        lo32, hi32 = np.float32(lo), np.float32(hi)  # the ABI takes floats: compare against what the kernel receives
        got = ctx.decode_count_range(dcol, float(lo32), float(hi32))
        ctx.synchronize()
        with np.errstate(invalid="ignore"):
            want = ((v >= lo32) & (v <= hi32)).sum(axis=1)
        assert np.array_equal(got.cpu().numpy().astype(np.int64), want), (name, lo, hi)
