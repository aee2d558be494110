"""GPU differential fuzzing against the oracle: random columns built to sit on the decision boundaries of the codec — magnitudes
around 2^51 / 2^53 / 2^63 scaled by powers of ten (the ranges where the encode kernel switches between its shortcut and the
literal arithmetic), mixed precisions inside a rowgroup, runs of constants, specials, exception bursts, ALP_RD stretches — whole
streams compared byte for byte, then decoded back.  Seeds are fixed; ALPGPU_FUZZ_ROUNDS=n runs more rounds, ALPGPU_FUZZ_SEED_BASE=k
shifts them to fresh seeds.  The last test does the same against the REAL reference (oracle/_ref), a quarter as many rounds."""
import os

import numpy as np
import pytest
import torch

import layout

pytestmark = pytest.mark.gpu
ROUNDS = int(os.environ.get("ALPGPU_FUZZ_ROUNDS", "24"))
SEED_BASE = int(os.environ.get("ALPGPU_FUZZ_SEED_BASE", "0"))


def fuzz_column(rng, dtype):
    n_vec = int(rng.integers(1, 260))
    n = n_vec * 1024
    f64 = dtype == np.float64
    out = np.empty(n, np.float64)
    pos = 0
    while pos < n:
        seg = int(min(n - pos, rng.integers(1, 40) * 1024 + int(rng.integers(0, 1024))))
        kind = rng.integers(0, 9)
        max_d = 15 if f64 else 6
        if kind == 0:  # decimals of one precision and magnitude
            d = int(rng.integers(0, max_d))
            mag = 10.0 ** rng.integers(-3, 12 if f64 else 5)
            x = np.round(rng.uniform(-mag, mag, seg), d)
        elif kind == 1:  # integers near the shortcut's range limits, divided by a power of ten
            e = int(rng.integers(0, 19 if f64 else 10))
            lim = float(2 ** int(rng.choice([51, 52, 53, 62, 63] if f64 else [22, 23, 24, 30, 31])))
            k = np.floor(rng.uniform(0.5, 1.5, seg) * lim) * rng.choice([-1.0, 1.0], seg)
            x = k / 10.0 ** e
        elif kind == 2:  # constants and short runs
            vals = np.round(rng.uniform(-1e4, 1e4, 8), int(rng.integers(0, 4)))
            x = np.repeat(vals[rng.integers(0, 8, (seg + 63) // 64)], 64)[:seg]
        elif kind == 3:  # full-precision values: ALP_RD material
            x = rng.uniform(-1.0, 1.0, seg) * 10.0 ** rng.integers(-5, 6)
        elif kind == 4:  # mixed precisions value by value
            d = rng.integers(0, max_d, seg)
            x = np.round(rng.uniform(0, 1000, seg) * 10.0 ** d) / 10.0 ** d
        elif kind == 5:  # tiny and huge
            x = rng.uniform(-1, 1, seg) * 10.0 ** rng.integers(-300 if f64 else -40, 300 if f64 else 38, seg).astype(np.float64)
        elif kind == 6:  # exception bursts inside clean decimals
            x = np.round(rng.uniform(0, 100, seg), 2)
            for _ in range(int(rng.integers(1, 6))):
                a = int(rng.integers(0, seg))
                b = min(seg, a + int(rng.integers(1, 400)))
                x[a:b] = rng.uniform(0, 100, b - a) * np.pi
        elif kind == 7:  # specials sprinkled
            x = np.round(rng.uniform(-50, 50, seg), 1)
            m = rng.random(seg) < rng.choice([0.001, 0.02, 0.3])
            x[m] = rng.choice(np.array([np.nan, np.inf, -np.inf, -0.0, 0.0]), int(m.sum()))
        else:  # zeros and sign flips
            x = np.where(rng.random(seg) < 0.5, 0.0, np.round(rng.uniform(-1, 1, seg), 3))
        out[pos:pos + seg] = x
        pos += seg
    with np.errstate(over="ignore", invalid="ignore"):
        return out.astype(dtype)


def _sinks_agree(ctx, dcol):
    """the fused consumers' two kernels (one wavefront per vector, words straight from HBM / four wavefronts over an LDS stage) document one
    summation order: same bits (NaN where NaN), same counts, on whatever the fuzz column holds"""
    from alp_amd import capi
    got = {}
    for mode in (2, 3):
        ctx.set_option(capi.OPT_CONSUMER_PIPELINED, mode)
        sums = ctx.decode_sum(dcol)
        cnts = ctx.decode_count_range(dcol, -1.0e3, 1.0e3)
        ctx.synchronize()
        got[mode] = (sums.cpu().numpy(), cnts.cpu().numpy())
    ctx.set_option(capi.OPT_CONSUMER_PIPELINED, 0)
    a, b = got[2][0], got[3][0]
    same = (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))
    assert same.all(), f"sums differ between the two kernels at vectors {np.nonzero(~same)[0][:5]}"
    assert np.array_equal(got[2][1], got[3][1]), "counts differ between the two kernels"


@pytest.mark.parametrize("seed", list(range(ROUNDS)))
def test_fuzz_double(ctx, oracle, seed):
    from alp_amd import capi
    col_np = fuzz_column(np.random.default_rng(7000 + SEED_BASE + seed), np.float64)
    want = layout.compact(oracle.encode_column(col_np))
    x = torch.from_numpy(col_np).cuda()
    dcol = capi.DeviceColumn(col_np.size // 1024)
    ctx.encode(x, dcol)
    ctx.synchronize()
    assert ctx.column_totals(dcol)[2] == 0
    for a, b, what in zip(dcol.to_host(), want, ("rowgroup states", "descriptors", "packed stream", "exception stream")):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"seed {seed}: {what}"
    out = ctx.decode(dcol)
    ctx.synchronize()
    assert torch.equal(out.view(torch.int64), x.view(torch.int64))
    out2 = ctx.decode(capi.DeviceColumn.from_host(*want))  # the decoder on the oracle's own encoding
    ctx.synchronize()
    assert torch.equal(out2.view(torch.int64), x.view(torch.int64))
    _sinks_agree(ctx, dcol)


@pytest.mark.parametrize("seed", list(range(ROUNDS)))
def test_fuzz_float(ctx, seed):
    from alp_amd import capi
    from oracle.pyoracle import OracleF32
    col_np = fuzz_column(np.random.default_rng(9000 + SEED_BASE + seed), np.float32)
    want = layout.compact(OracleF32().encode_column(col_np), 4)
    x = torch.from_numpy(col_np).cuda()
    dcol = capi.DeviceColumn(col_np.size // 1024, dtype="f32")
    ctx.encode(x, dcol)
    ctx.synchronize()
    assert ctx.column_totals(dcol)[2] == 0
    for a, b, what in zip(dcol.to_host(), want, ("rowgroup states", "descriptors", "packed stream", "exception stream")):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"seed {seed}: {what}"
    out = ctx.decode(dcol)
    ctx.synchronize()
    assert torch.equal(out.view(torch.int32), x.view(torch.int32))
    out2 = ctx.decode(capi.DeviceColumn.from_host(*want, dtype="f32"))
    ctx.synchronize()
    assert torch.equal(out2.view(torch.int32), x.view(torch.int32))
    _sinks_agree(ctx, dcol)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("seed", list(range(max(1, ROUNDS // 4))))
def test_fuzz_against_the_reference(ctx, ref, seed, dtype):
    """no restatement in between: the GPU streams against the streams of the reference compiled in place (oracle/_ref)"""
    from alp_amd import capi
    from oracle.pyoracle import ReferenceF32
    if dtype == "f32" and not ReferenceF32.available():
        pytest.skip("oracle/_ref without float entry points")
    f64 = dtype == "f64"
    col_np = fuzz_column(np.random.default_rng(11000 + SEED_BASE + seed), np.float64 if f64 else np.float32)
    want = layout.compact(ref.encode_column(col_np)) if f64 else layout.compact(ReferenceF32().encode_column(col_np), 4)
    x = torch.from_numpy(col_np).cuda()
    dcol = capi.DeviceColumn(col_np.size // 1024, dtype=dtype)
    ctx.encode(x, dcol)
    ctx.synchronize()
    assert ctx.column_totals(dcol)[2] == 0
    for a, b, what in zip(dcol.to_host(), want, ("rowgroup states", "descriptors", "packed stream", "exception stream")):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"seed {seed} {dtype}: {what} differ from the reference"
    out = ctx.decode(dcol)
    ctx.synchronize()
    it = torch.int64 if f64 else torch.int32
    assert torch.equal(out.view(it), x.view(it))
