"""CPU: the plain-C restatement (oracle/alp_oracle.c) against the committed golden vectors produced by the
real reference, and against the (bit width, exception count) pairs the reference's own unit test asserts
(test/test_alp_sample.cpp:178-179)."""
import numpy as np
import pytest

import golden_io

FIRST = golden_io.first_vectors()
RG = golden_io.rowgroup_samples()


@pytest.mark.parametrize("name,col,gold,known", FIRST, ids=[c[0] for c in FIRST])
def test_first_vector_matches_reference(oracle, name, col, gold, known):
    got = oracle.encode_column(col)
    golden_io.assert_same_encoding(got, gold, name)
    if gold["scheme"][0] == 2 and known[0] >= 0:  # the reference asserts these only on the ALP branch
        assert int(got["bw"][0]) == int(known[0]) and int(got["exc_cnt"][0]) == int(known[1])
    dec = oracle.decode_column(got)
    assert np.array_equal(dec.view(np.uint64), col.view(np.uint64)), "decode(encode(x)) must reproduce every bit"


@pytest.mark.parametrize("name,col,gold", RG, ids=[c[0] for c in RG])
def test_rowgroup_sample_matches_reference(oracle, name, col, gold):
    got = oracle.encode_column(col)
    golden_io.assert_same_encoding(got, gold, name)
    dec = oracle.decode_column(got)
    assert np.array_equal(dec.view(np.uint64), col.view(np.uint64))


def test_known_multi_vector_statistics():
    # SURVEY.md §4 (verified against the reference build): sum of bit widths / exceptions over 128 vectors
    want = {"bitcoin_transactions_f_tw": (3847, 10870), "city_temperature_f_tw": (1359, 193), "food_prices_tw": (2083, 4129),
            "gov26_tw": (35, 818), "nyc29_tw": (5096, 1044)}
    for name, _, gold in RG:
        assert (int(gold["bw"].sum()), int(gold["exc_cnt"].sum())) == want[name]


# ---- single precision -------------------------------------------------------------------------------------------
FLOATS = golden_io.float_vectors()


@pytest.mark.parametrize("name,col,gold,known", FLOATS, ids=[c[0] for c in FLOATS])
def test_float_column_matches_reference(name, col, gold, known):
    """oracle/alp_oracle_f32.c against the reference's float outputs; the first five are the reference's own float
    test columns with the bit widths / exception count its unit test asserts (data/include/float/test.hpp:10-14,
    float/edge_case.hpp:10)"""
    from oracle.pyoracle import OracleF32
    o = OracleF32()
    got = o.encode_column(col)
    golden_io.assert_same_encoding(got, gold, name, word=np.uint32)
    if known[0] >= 0:
        assert got["scheme"][0] == 2 and int(got["bw"][0]) == int(known[0])
    if known[1] >= 0:
        assert int(got["exc_cnt"][0]) == int(known[1])
    assert np.array_equal(o.decode_column(got).view(np.uint32), col.view(np.uint32))
