"""GPU against the REAL reference, directly: where oracle/_ref/libalp_ref.so (the reference compiled in place in the build
container, shipped with the snapshot) is present on the GPU box, the GPU encoder's streams are compared with the reference's
own outputs without the C restatement in between.  Double and float, ALP and ALP_RD rowgroups."""
import numpy as np
import pytest
import torch

import datagen
import layout

pytestmark = pytest.mark.gpu


def _gpu_streams(ctx, col_np, dtype):
    from alp_amd import capi
    x = torch.from_numpy(np.ascontiguousarray(col_np)).cuda()
    col = capi.DeviceColumn(col_np.size // 1024, dtype=dtype)
    ctx.encode(x, col)
    ctx.synchronize()
    assert ctx.column_totals(col)[2] == 0
    out = ctx.decode(col)
    ctx.synchronize()
    assert torch.equal(out.view(torch.int64 if dtype == "f64" else torch.int32), x.view(torch.int64 if dtype == "f64" else torch.int32))
    return col.to_host()


DOUBLE_COLUMNS = {
    "mixed": lambda: datagen.mixed_column(230, seed=41, exc_rate=0.02),
    "drifting": lambda: datagen.drifting_column(150, seed=42),
    "rd_unit": lambda: datagen.rd_column(140, seed=43),
    "alp_then_rd": lambda: np.concatenate([datagen.decimal_column(100, 2, seed=44), datagen.rd_column(120, seed=45, kind="latlon")]),
    "adversarial": lambda: np.concatenate(list(datagen.adversarial_vectors().values())),
}
FLOAT_COLUMNS = {
    "mixed": lambda: datagen.mixed_column_f32(230, seed=51, exc_rate=0.02),
    "drifting": lambda: datagen.drifting_column_f32(150, seed=52),
    "rd_unit": lambda: datagen.rd_column_f32(140, seed=53),
    "adversarial": lambda: np.concatenate(list(datagen.adversarial_vectors_f32().values())),
}


@pytest.mark.parametrize("name", list(DOUBLE_COLUMNS.keys()))
def test_double_streams_equal_the_reference(ctx, ref, name):
    col_np = DOUBLE_COLUMNS[name]()
    want = layout.compact(ref.encode_column(col_np))
    got = _gpu_streams(ctx, col_np, "f64")
    for a, b, what in zip(got, want, ("rowgroup states", "vector descriptors", "packed stream", "exception stream")):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"{name}: {what} differ from the reference"


@pytest.mark.parametrize("name", list(FLOAT_COLUMNS.keys()))
def test_float_streams_equal_the_reference(ctx, name):
    from oracle.pyoracle import ReferenceF32
    if not ReferenceF32.available():
        pytest.skip("oracle/_ref without float entry points")
    col_np = FLOAT_COLUMNS[name]()
    want = layout.compact(ReferenceF32().encode_column(col_np), 4)
    got = _gpu_streams(ctx, col_np, "f32")
    for a, b, what in zip(got, want, ("rowgroup states", "vector descriptors", "packed stream", "exception stream")):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"{name}: {what} differ from the reference"
