"""CPU, build container only (skipped when oracle/_ref is absent): the C restatement against the REAL
reference on seeded random inputs — primitives for every bit width, and whole columns of every class."""
import numpy as np
import pytest

import datagen
import golden_io


@pytest.mark.parametrize("bw", list(range(0, 65)))
def test_ffor_unffor_u64_all_bit_widths(oracle, ref, bw):
    rng = np.random.default_rng(bw)
    base = int(rng.integers(-2**62, 2**62))
    span = (1 << bw) - 1 if bw < 64 else (1 << 64) - 1
    vals = (rng.integers(0, 2**63, 1024, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, 1024, dtype=np.uint64)) & np.uint64(span)
    vals = (vals + np.uint64(base & (2**64 - 1)))  # wraps
    a, b = oracle.ffor_u64(vals, bw, base), ref.ffor_u64(vals, bw, base)
    assert np.array_equal(a[:16 * bw], b[:16 * bw])
    ua, ub = oracle.unffor_u64(b, bw, base), ref.unffor_u64(b, bw, base)
    assert np.array_equal(ua, ub) and np.array_equal(ua, vals)
    if bw < 64:  # falp bw=64 is broken in the reference
        fa, fb = oracle.falp(b, bw, base, 3, 5), ref.falp(b, bw, base, 3, 5)
        assert np.array_equal(fa.view(np.uint64), fb.view(np.uint64))
    fu = ref.unffor_decode(b, bw, base, 3, 5)
    assert np.array_equal(oracle.falp(b, bw, base, 3, 5).view(np.uint64), fu.view(np.uint64))


@pytest.mark.parametrize("bw", list(range(0, 17)))
def test_ffor_unffor_u16_all_bit_widths(oracle, ref, bw):
    rng = np.random.default_rng(100 + bw)
    vals = (rng.integers(0, 1 << 16, 1024) & ((1 << bw) - 1)).astype(np.uint16)
    a, b = oracle.ffor_u16(vals, bw), ref.ffor_u16(vals, bw)
    assert np.array_equal(a[:64 * bw], b[:64 * bw])
    assert np.array_equal(oracle.unffor_u16(b, bw), ref.unffor_u16(b, bw))
    assert np.array_equal(oracle.unffor_u16(b, bw), vals)


@pytest.mark.parametrize("bw", list(range(0, 9)))
def test_ffor_unffor_u8_all_bit_widths(oracle, ref, bw):
    rng = np.random.default_rng(200 + bw)
    base = int(rng.integers(0, 256))
    vals = (((rng.integers(0, 256, 1024) & ((1 << bw) - 1)) + base) & 0xFF).astype(np.uint8)
    a, b = oracle.ffor_u8(vals, bw, base), ref.ffor_u8(vals, bw, base)
    assert np.array_equal(a[:128 * bw], b[:128 * bw])
    assert np.array_equal(oracle.unffor_u8(b, bw, base), ref.unffor_u8(b, bw, base))
    assert np.array_equal(oracle.unffor_u8(b, bw, base), vals)


@pytest.mark.parametrize("name", list(datagen.adversarial_vectors().keys()))
@pytest.mark.parametrize("ef", [(14, 12), (18, 18), (0, 0), (5, 2), (16, 0)])
def test_encode_simdized_corner_cases(oracle, ref, name, ef):
    vec = datagen.adversarial_vectors()[name]
    e, f = ef
    ea, xa, pa, ca = oracle.encode_simdized(vec, f, e)
    eb, xb, pb, cb = ref.encode_simdized(vec, f, e)
    assert ca == cb and np.array_equal(ea, eb)
    assert np.array_equal(pa[:ca], pb[:cb]) and np.array_equal(xa[:ca].view(np.uint64), xb[:cb].view(np.uint64))
    assert oracle.analyze_ffor(ea) == ref.analyze_ffor(eb)


COLUMNS = {
    "decimal2": lambda: datagen.decimal_column(230, 2, seed=1),
    "decimal5_small": lambda: datagen.decimal_column(101, 5, 0, 10, seed=2),
    "mixed_1pct": lambda: datagen.mixed_column(250, seed=3, exc_rate=0.01),
    "mixed_10pct": lambda: datagen.mixed_column(120, seed=4, exc_rate=0.10),
    "rd_unit": lambda: datagen.rd_column(130, seed=5, kind="unit"),
    "rd_latlon": lambda: datagen.rd_column(110, seed=6, kind="latlon"),
    "drifting_k": lambda: datagen.drifting_column(200, seed=7),
    "integers": lambda: np.floor(datagen.decimal_column(64, 0, 0, 1e6, seed=8)),
    "tiny": lambda: datagen.decimal_column(3, 3, seed=9),
    "rd_few_left_parts": lambda: (np.random.default_rng(10).integers(0, 5, 150 * 1024).astype(np.float64) * 1e-3
                                  + np.random.default_rng(11).random(150 * 1024) * 1e-9),
}


@pytest.mark.parametrize("name", list(COLUMNS.keys()))
def test_whole_column_matches_reference(oracle, ref, name):
    col = COLUMNS[name]()
    a, b = oracle.encode_column(col), ref.encode_column(col)
    golden_io.assert_same_encoding(a, b, name)
    dec = oracle.decode_column(a)
    assert np.array_equal(dec.view(np.uint64), col.view(np.uint64))


@pytest.mark.parametrize("seed", list(range(40)))
def test_fuzz_columns_oracle_equals_reference(oracle, ref, seed):
    """the boundary-hunting random columns of tests/test_fuzz_gpu.py, restatement against the real reference (both precisions)"""
    from oracle.pyoracle import OracleF32, ReferenceF32
    from test_fuzz_gpu import fuzz_column
    c = fuzz_column(np.random.default_rng(7000 + seed), np.float64)
    a, b = oracle.encode_column(c), ref.encode_column(c)
    golden_io.assert_same_encoding(a, b, f"f64 seed {seed}")
    assert np.array_equal(oracle.decode_column(a).view(np.uint64), c.view(np.uint64))
    if ReferenceF32.available():
        of, rf = OracleF32(), ReferenceF32()
        c = fuzz_column(np.random.default_rng(9000 + seed), np.float32)
        a, b = of.encode_column(c), rf.encode_column(c)
        golden_io.assert_same_encoding(a, b, f"f32 seed {seed}", word=np.uint32)
        assert np.array_equal(of.decode_column(a).view(np.uint32), c.view(np.uint32))


def _unique_bits(values):
    return values[np.unique(values.view(np.uint64), return_index=True)[1]]


def test_search_boundary_values_oracle_equals_reference(oracle, ref):
    """one constant vector per boundary value of the (e,f) search (datagen.search_boundary_values): the rowgroup state shows
    the first candidate that round-trips it; restatement against the real reference"""
    vals = _unique_bits(datagen.search_boundary_values())
    for i, v in enumerate(vals):
        col = np.full(1024, v)
        a, b = oracle.encode_column(col), ref.encode_column(col)
        golden_io.assert_same_encoding(a, b, f"boundary value {i}: {v!r}")


@pytest.mark.parametrize("seed", list(range(30)))
def test_search_boundary_mixtures_oracle_equals_reference(oracle, ref, seed):
    col = datagen.search_boundary_mixtures(seed)
    a, b = oracle.encode_column(col), ref.encode_column(col)
    golden_io.assert_same_encoding(a, b, f"boundary mixture {seed}")
    assert np.array_equal(oracle.decode_column(a).view(np.uint64), col.view(np.uint64))
