"""CPU, build container only (skipped when oracle/_ref is absent): the float restatement (oracle/alp_oracle_f32.c)
against the REAL reference's float instantiation — scalar encode/decode, primitives for every bit width, corner-case
vectors, whole columns — plus the reference's own float test columns with the answers its unit test asserts."""
import itertools

import numpy as np
import pytest

import datagen
import golden_io


@pytest.fixture(scope="module")
def of32():
    from oracle.pyoracle import OracleF32
    return OracleF32()


@pytest.fixture(scope="module")
def rf32():
    from oracle.pyoracle import ReferenceF32
    if not ReferenceF32.available():
        pytest.skip("oracle/_ref without float entry points")
    return ReferenceF32()


SCALARS = [0.0, -0.0, 1.5, -1.5, 2.5, 0.5, 12345.678, 1e-3, 1e-10, 1e-45, 1.1754942e-38, 4194303.5, 8388609.0, 16777216.0,
           2147483520.0, 2147483648.0, -2147483648.0, -2147483904.0, 3e9, -3e9, 9.2e18, 1e30, float("inf"), float("-inf"), float("nan")]


def test_scalar_encode_decode_value(of32, rf32):
    for v, e in itertools.product(SCALARS, range(11)):
        for f in range(e + 1):
            a, b = of32.encode_value(v, f, e), rf32.encode_value(v, f, e)
            assert a == b, (v, e, f, a, b)
    rng = np.random.default_rng(0)
    for enc in list(rng.integers(-2**31, 2**31, 200)) + [0, 1, -1, 2**31 - 1, -2**31]:
        for e in range(11):
            for f in range(min(e, 9) + 1):  # f = 10 reads past FACT_ARR in the reference (oracle header, U4)
                a, b = of32.decode_value(int(enc), f, e), rf32.decode_value(int(enc), f, e)
                assert np.float32(a).view(np.uint32) == np.float32(b).view(np.uint32), (enc, e, f)


@pytest.mark.parametrize("bw", list(range(0, 33)))
def test_ffor_unffor_falp_u32_all_bit_widths(of32, rf32, bw):
    rng = np.random.default_rng(bw)
    base = int(rng.integers(-2**30, 2**30))
    span = (1 << bw) - 1
    vals = (rng.integers(0, 2**32, 1024, dtype=np.uint64) & np.uint64(span)).astype(np.uint32)
    vals = (vals + np.uint32(base & 0xFFFFFFFF))  # wraps
    a, b = of32.ffor_u32(vals, bw, base), rf32.ffor_u32(vals, bw, base)
    assert np.array_equal(a[:32 * bw], b[:32 * bw])
    ua, ub = of32.unffor_u32(b, bw, base), rf32.unffor_u32(b, bw, base)
    assert np.array_equal(ua, ub) and np.array_equal(ua, vals)
    for (e, f) in [(5, 3), (0, 0), (10, 9), (7, 0)]:
        fa = of32.falp(b, bw, base, f, e)
        assert np.array_equal(fa.view(np.uint32), rf32.unffor_decode(b, bw, base, f, e).view(np.uint32))
        if bw < 32:  # the fused kernel's widest case mirrors the double path's H5 defect; see test below
            assert np.array_equal(fa.view(np.uint32), rf32.falp(b, bw, base, f, e).view(np.uint32)), (bw, e, f)


def test_falp_bw32_reference_behaviour_is_recorded(of32, rf32):
    """documents whether the reference's fused 32-bit falp agrees with unffor+decode at bw = 32"""
    rng = np.random.default_rng(99)
    vals = rng.integers(0, 2**32, 1024, dtype=np.uint64).astype(np.uint32)
    b = rf32.ffor_u32(vals, 32, 0)
    fused, unfused = rf32.falp(b, 32, 0, 0, 0), rf32.unffor_decode(b, 32, 0, 0, 0)
    assert np.array_equal(of32.falp(b, 32, 0, 0, 0).view(np.uint32), unfused.view(np.uint32))
    print("reference falp(float, bw=32) == unffor+decode:", np.array_equal(fused.view(np.uint32), unfused.view(np.uint32)))


@pytest.mark.parametrize("name", list(datagen.adversarial_vectors_f32().keys()))
@pytest.mark.parametrize("ef", [(2, 0), (10, 10), (0, 0), (5, 2), (10, 0), (9, 9), (7, 4)])
def test_encode_simdized_corner_cases(of32, rf32, name, ef):
    vec = datagen.adversarial_vectors_f32()[name]
    e, f = ef
    ea, xa, pa, ca = of32.encode_simdized(vec, f, e)
    eb, xb, pb, cb = rf32.encode_simdized(vec, f, e)
    assert ca == cb and np.array_equal(ea, eb)
    assert np.array_equal(pa[:ca], pb[:cb]) and np.array_equal(xa[:ca].view(np.uint32), xb[:cb].view(np.uint32))
    assert of32.analyze_ffor(ea) == rf32.analyze_ffor(eb)


COLUMNS = {
    "decimal2": lambda: datagen.decimal_column_f32(230, 2, seed=1),
    "decimal1_wide": lambda: datagen.decimal_column_f32(120, 1, -1e5, 1e5, seed=12),
    "decimal4_small": lambda: datagen.decimal_column_f32(101, 4, 0, 10, seed=2),
    "mixed_1pct": lambda: datagen.mixed_column_f32(250, seed=3, exc_rate=0.01),
    "mixed_10pct": lambda: datagen.mixed_column_f32(120, seed=4, exc_rate=0.10),
    "rd_unit": lambda: datagen.rd_column_f32(130, seed=5, kind="unit"),
    "rd_latlon": lambda: datagen.rd_column_f32(110, seed=6, kind="latlon"),
    "drifting_k": lambda: datagen.drifting_column_f32(200, seed=7),
    "integers": lambda: np.floor(datagen.decimal_column_f32(64, 0, 0, 1e6, seed=8)),
    "tiny": lambda: datagen.decimal_column_f32(3, 3, seed=9),
    "adversarial": lambda: np.concatenate(list(datagen.adversarial_vectors_f32().values())),
    "negzero_samples": lambda: np.where(np.random.default_rng(13).random(150 * 1024) < 0.3, np.float32(-0.0),
                                        datagen.decimal_column_f32(150, 2, seed=14)).astype(np.float32),
    "rd_few_left_parts": lambda: (np.random.default_rng(10).integers(0, 5, 150 * 1024) * 1e-3
                                  + np.random.default_rng(11).random(150 * 1024) * 1e-6).astype(np.float32),
}


@pytest.mark.parametrize("name", list(COLUMNS.keys()))
def test_whole_column_matches_reference(of32, rf32, name):
    col = COLUMNS[name]()
    a, b = of32.encode_column(col), rf32.encode_column(col)
    golden_io.assert_same_encoding(a, b, name, word=np.uint32)
    dec = of32.decode_column(a)
    assert np.array_equal(dec.view(np.uint32), col.view(np.uint32))


def test_reference_float_test_columns(of32, rf32):
    """data/float/test_{0..3}.csv and data/edge_case/avx512dq.csv with the values the reference's unit test asserts
    (data/include/float/test.hpp:10-14: bit widths 4, 10, 17, 0; float/edge_case.hpp:10: 192 exceptions, bw 0)"""
    import os
    root = "/root/reference/data"
    if not os.path.isdir(root):
        pytest.skip("reference data absent")
    expect = {"float/test_0.csv": (4, None), "float/test_1.csv": (10, None), "float/test_2.csv": (17, None),
              "float/test_3.csv": (0, None), "edge_case/avx512dq.csv": (0, 192)}
    for rel, (bw, exc) in expect.items():
        vals = [np.float32(l.strip().rstrip(",")) for l in open(os.path.join(root, rel)) if l.strip()]
        col = np.array(vals + [0.0] * (1024 - len(vals)), np.float32)
        a, b = of32.encode_column(col), rf32.encode_column(col)
        golden_io.assert_same_encoding(a, b, rel, word=np.uint32)
        assert a["scheme"][0] == 2 and int(a["bw"][0]) == bw, rel
        if exc is not None:
            assert int(a["exc_cnt"][0]) == exc
        assert np.array_equal(of32.decode_column(a).view(np.uint32), col.view(np.uint32))
