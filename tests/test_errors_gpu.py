"""GPU: error behaviour of the C ABI — capacity overflow is reported (and nothing is written past the buffers), bad
arguments are rejected with ALPGPU_ERR_INVALID, empty columns are fine."""
import ctypes as C

import numpy as np
import pytest
import torch

import datagen

pytestmark = pytest.mark.gpu


def test_capacity_overflow_is_reported_not_written(ctx):
    from alp_amd import capi
    col_np = datagen.rd_column(40, seed=3)  # ~7 KiB of packed words per vector
    x = torch.from_numpy(col_np).cuda()
    guard = 4096
    col = capi.DeviceColumn(40, packed_capacity=40 * 1024 + guard, exc_capacity=64 + guard)
    col.packed.fill_(0xAB)
    col.c.packed_capacity = 40 * 1024  # the library believes the stream ends here; the guard zone must stay intact
    lib = capi.lib
    assert lib.alpgpu_encode_f64(ctx.h, C.c_void_p(x.data_ptr()), 40, C.byref(col.c)) == 0
    pb, eb, ov = C.c_uint64(), C.c_uint64(), C.c_int()
    rc = lib.alpgpu_column_totals(ctx.h, C.byref(col.c), C.byref(pb), C.byref(eb), C.byref(ov))
    assert rc == -4 and ov.value == 1 and pb.value > 40 * 1024, "overflow must be reported with the needed size"
    assert bool((col.packed[40 * 1024:] == 0xAB).all()), "nothing may be written past the stream's capacity"
    # retry with the reported size succeeds and round-trips
    col2 = capi.DeviceColumn(40, packed_capacity=int(pb.value) + 1024, exc_capacity=int(eb.value) + 64)
    ctx.encode(x, col2)
    out = ctx.decode(col2)
    ctx.synchronize()
    assert ctx.column_totals(col2)[2] == 0
    assert torch.equal(out.view(torch.int64), x.view(torch.int64))


def test_invalid_arguments_are_rejected(ctx):
    from alp_amd import capi
    lib = capi.lib
    col = capi.DeviceColumn(10)
    x = torch.zeros(10 * 1024, dtype=torch.float64, device="cuda")
    assert lib.alpgpu_encode_f64(ctx.h, C.c_void_p(x.data_ptr()), 11, C.byref(col.c)) == -2, "n_vectors must match the column"
    assert lib.alpgpu_encode_f64(ctx.h, C.c_void_p(0), 10, C.byref(col.c)) == -2
    assert lib.alpgpu_decode_f64(ctx.h, C.byref(col.c), C.c_void_p(0)) == -2
    assert lib.alpgpu_set_option(ctx.h, 1, 3) == -2 and lib.alpgpu_set_option(ctx.h, 1, -1) == -2 and lib.alpgpu_set_option(ctx.h, 99, 0) == -2
    assert b"" != lib.alpgpu_last_error()
    h = C.c_void_p()
    assert lib.alpgpu_ctx_create(99, C.byref(h)) == -2 and not h.value


def test_empty_column(ctx):
    from alp_amd import capi
    col = capi.DeviceColumn(0)
    x = torch.zeros(0, dtype=torch.float64, device="cuda")
    ctx.encode(x, col)
    out = ctx.decode(col)
    ctx.synchronize()
    assert out.numel() == 0 and ctx.column_totals(col) == (0, 0, 0)
