"""GPU: the encoder heals itself.  The single-pass encode needs its predecessor workgroups to run (HIP promises no dispatch
order); when its look-back gives up, the two-pass kernels enqueued behind it on the same stream redo the column.  The debug
option ALPGPU_OPT_DEBUG_FORCE_STALL makes every look-back that has to wait give up at once; the caller must still get a complete
column, byte for byte the oracle's.  Also here: the two-pass form for float columns, encodes on switching streams (one shared
workspace), columns that overflowed (a decoder must stay inside the buffers), empty float columns, short blobs."""
import ctypes as C

import numpy as np
import pytest
import torch

import datagen
import layout

pytestmark = pytest.mark.gpu


def _streams_equal(dcol, want, value_bytes):
    for a, b, what in zip(dcol.to_host(), layout.compact(want, value_bytes), ("rowgroup states", "descriptors", "packed stream", "exception stream")):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), what


@pytest.fixture()
def stalling_ctx(ctx):
    from alp_amd import capi
    ctx.set_option(capi.OPT_DEBUG_FORCE_STALL, 1)
    yield ctx
    ctx.set_option(capi.OPT_DEBUG_FORCE_STALL, 0)


def test_forced_stall_is_recovered_double(stalling_ctx, oracle):
    from alp_amd import capi
    ctx = stalling_ctx
    col_np = np.concatenate([datagen.mixed_column(330, seed=31, exc_rate=0.02), datagen.rd_column(170, seed=32), datagen.drifting_column(200, seed=33)])
    want = oracle.encode_column(col_np)
    x = torch.from_numpy(col_np).cuda()
    dcol = capi.DeviceColumn(col_np.size // 1024)
    dcol.packed.fill_(0x5A)  # a stalled single pass leaves its tiles unwritten: the recovery must write every byte
    dcol.exc.fill_(0xA5)
    ctx.encode(x, dcol)
    assert ctx.column_totals(dcol)[2] == 0, "no overflow, and no stall error either (column_totals raises on one)"
    assert int(dcol.totals[6]) == 1, "the test is vacuous unless the single pass really gave up"
    _streams_equal(dcol, want, 8)
    out = ctx.decode(dcol)
    ctx.synchronize()
    assert torch.equal(out.view(torch.int64), x.view(torch.int64))


def test_forced_stall_is_recovered_float(stalling_ctx):
    from alp_amd import capi
    from oracle.pyoracle import OracleF32
    ctx = stalling_ctx
    col_np = np.concatenate([datagen.mixed_column_f32(310, seed=41, exc_rate=0.02), datagen.rd_column_f32(190, seed=42)])
    want = OracleF32().encode_column(col_np)
    x = torch.from_numpy(col_np).cuda()
    dcol = capi.DeviceColumn(col_np.size // 1024, dtype="f32")
    dcol.packed.fill_(0x5A)
    dcol.exc.fill_(0xA5)
    ctx.encode(x, dcol)
    assert ctx.column_totals(dcol)[2] == 0 and int(dcol.totals[6]) == 1
    _streams_equal(dcol, want, 4)
    out = ctx.decode(dcol)
    ctx.synchronize()
    assert torch.equal(out.view(torch.int32), x.view(torch.int32))


def test_no_stall_no_recovery(ctx):
    from alp_amd import capi
    x = torch.from_numpy(datagen.mixed_column(256, seed=3)).cuda()
    dcol = ctx.encode(x)
    ctx.synchronize()
    assert int(dcol.totals[3]) == 0 and int(dcol.totals[6]) == 0


def test_two_pass_option_applies_to_float_columns(ctx):
    from alp_amd import capi
    from oracle.pyoracle import OracleF32
    col_np = np.concatenate([datagen.mixed_column_f32(1300, seed=51, exc_rate=0.03), datagen.rd_column_f32(200, seed=52)])  # > one scan tile of 1024 vectors
    want = OracleF32().encode_column(col_np)
    x = torch.from_numpy(col_np).cuda()
    ctx.set_option(capi.OPT_ENCODE_TWO_PASS, 1)
    try:
        dcol = capi.DeviceColumn(col_np.size // 1024, dtype="f32")
        dcol.packed.fill_(0x11)
        dcol.exc.fill_(0x22)
        ctx.encode(x, dcol)
        assert ctx.column_totals(dcol)[2] == 0
    finally:
        ctx.set_option(capi.OPT_ENCODE_TWO_PASS, 0)
    _streams_equal(dcol, want, 4)
    # and a float capacity overflow in the two-pass form is reported, not written
    small = capi.DeviceColumn(col_np.size // 1024, dtype="f32", packed_capacity=64 * 1024, exc_capacity=4096)
    ctx.set_option(capi.OPT_ENCODE_TWO_PASS, 1)
    try:
        ctx.encode(x, small)
        with pytest.raises(capi.AlpGpuError):
            ctx.column_totals(small)
    finally:
        ctx.set_option(capi.OPT_ENCODE_TWO_PASS, 0)
    out = ctx.decode(small)  # unspecified values, but inside the buffers
    ctx.synchronize()
    assert out.numel() == col_np.size


def test_encodes_on_alternating_streams_share_the_workspace_safely(ctx, oracle):
    """one scan/status workspace per context; two encodes in flight on different streams used to race on it."""
    from alp_amd import capi
    cols = [datagen.mixed_column(2048, seed=61 + i, exc_rate=0.01) for i in range(2)]
    wants = [oracle.encode_column(c) for c in cols]
    xs = [torch.from_numpy(c).cuda() for c in cols]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for rep in range(4):
        dcols = []
        for i in (0, 1, 0, 1):
            with torch.cuda.stream(streams[i]):  # the context follows torch's current stream
                dcols.append((i, ctx.encode(xs[i])))
        torch.cuda.synchronize()
        for i, d in dcols:
            _streams_equal(d, wants[i], 8)


def test_decode_of_an_overflowed_column_stays_inside_the_buffers(ctx):
    from alp_amd import capi
    col_np = datagen.rd_column(300, seed=9)
    x = torch.from_numpy(col_np).cuda()
    guard = 1 << 16
    col = capi.DeviceColumn(300, packed_capacity=200 * 1024 + guard, exc_capacity=4096)
    col.c.packed_capacity = 200 * 1024
    col.vectors.fill_(0xFF)  # garbage descriptors: every one of them must be overwritten
    ctx.encode(x, col)
    with pytest.raises(capi.AlpGpuError):
        ctx.column_totals(col)
    vec = col.vectors.cpu().numpy().view(capi.VECTOR_DTYPE)[:300]
    ends = vec["packed_off"].astype(np.int64) + 128 * (vec["bw"].astype(np.int64) + vec["lbw"])
    assert (ends <= 200 * 1024).all(), "no descriptor may point past the stream"
    out = ctx.decode(col)
    ctx.synchronize()
    assert out.numel() == 300 * 1024


def test_empty_float_column_and_short_blob(ctx):
    from alp_amd import capi
    col = capi.DeviceColumn(0, dtype="f32")
    x = torch.zeros(0, dtype=torch.float32, device="cuda")
    ctx.encode(x, col)
    assert ctx.decode(col).numel() == 0 and ctx.column_totals(col) == (0, 0, 0)
    # an empty column with no buffers at all, straight through the C ABI
    empty = capi.CColumn(0, 0, None, None, None, 0, None, 0, None, 0, 0, None)
    lib = capi.lib
    assert lib.alpgpu_encode_f64(ctx.h, C.c_void_p(0), 0, C.byref(empty)) == 0
    pb, eb, ov = C.c_uint64(7), C.c_uint64(7), C.c_int(7)
    assert lib.alpgpu_column_totals(ctx.h, C.byref(empty), C.byref(pb), C.byref(eb), C.byref(ov)) == 0 and (pb.value, eb.value, ov.value) == (0, 0, 0)
    assert lib.alpgpu_decode_f64(ctx.h, C.byref(empty), C.c_void_p(0)) == 0
    with pytest.raises(capi.AlpGpuError):
        ctx.from_blob(np.zeros(40, np.uint8))
