"""GPU parity: rowgroup init + vector encode (HIP, through the C ABI) against the oracle, bit for bit:
rowgroup state (scheme, k, (e,f) candidates), per-vector (e, f, bit width, base, exception count), packed
bytes, exception values and positions, stream offsets; then encode->decode round trips on the GPU."""
import numpy as np
import pytest
import torch

import datagen
import golden_io
import layout

pytestmark = pytest.mark.gpu


def gpu_encode(ctx, col_np, states=None):
    from alp_amd import capi
    x = torch.from_numpy(np.ascontiguousarray(col_np)).cuda()
    col = capi.DeviceColumn(col_np.size // 1024)
    if states is None:
        ctx.encode(x, col)
    else:
        col.rowgroups[: states.size * 32] = torch.from_numpy(states.view(np.uint8).reshape(-1)).cuda()
        ctx.encode_vectors(x, col)
    ctx.synchronize()
    pb, eb, ov = ctx.column_totals(col)
    assert ov == 0
    return col, x


def assert_parts_equal(got, want, name):
    """got/want: fixed-stride dicts (layout.expand / oracle).  Per-vector metadata, packed u64 words (ALP words /
    ALP_RD right parts), exception values and positions must match in every bit."""
    n = want["scheme"].size
    for k in ("scheme", "e", "f", "bw", "lbw", "base", "exc_cnt"):
        assert np.array_equal(got[k], want[k]), f"{name}: {k} differs at {np.nonzero(got[k] != want[k])[0][:5]}"
    assert np.array_equal(got["packed"], want["packed"]), f"{name}: packed words differ"
    for v in range(n):
        c = int(want["exc_cnt"][v])
        assert np.array_equal(got["pos"][v, :c], want["pos"][v, :c]), f"{name}: exception positions differ in vector {v}"
        if want["scheme"][v] == 2:
            assert np.array_equal(got["exc"][v].view(np.uint64)[:c], want["exc"][v].view(np.uint64)[:c]), f"{name}: exc values v{v}"
        else:
            assert np.array_equal(got["exc"][v].view(np.uint16)[:c], want["exc"][v].view(np.uint16)[:c]), f"{name}: rd exc values v{v}"


COLUMNS = {
    "decimal2": lambda: datagen.decimal_column(230, 2, seed=1),
    "decimal5_small": lambda: datagen.decimal_column(101, 5, 0, 10, seed=2),
    "mixed_1pct": lambda: datagen.mixed_column(250, seed=3, exc_rate=0.01),
    "mixed_10pct": lambda: datagen.mixed_column(120, seed=4, exc_rate=0.10),
    "drifting_k": lambda: datagen.drifting_column(200, seed=7),
    "integers": lambda: np.floor(datagen.decimal_column(64, 0, 0, 1e6, seed=8)),
    "tiny": lambda: datagen.decimal_column(3, 3, seed=9),
    "adversarial": lambda: np.concatenate(list(datagen.adversarial_vectors().values())),
}


@pytest.mark.parametrize("name", list(COLUMNS.keys()))
def test_alp_columns_encode_bit_exact(ctx, oracle, name):
    col_np = COLUMNS[name]()
    want = oracle.encode_column(col_np)
    dcol, x = gpu_encode(ctx, col_np)
    rg, vec, packed, exc = dcol.to_host()
    got = layout.expand(rg, vec, packed, exc)
    assert np.array_equal(got["k"], want["k"]), name
    assert np.array_equal(got["combos"], want["combos"]), name
    assert_parts_equal(got, want, name)
    # offsets are the exclusive scan of the record sizes, in vector order
    w_rg, w_vec, w_packed, w_exc = layout.compact(want)
    assert np.array_equal(vec["packed_off"], w_vec["packed_off"]) and np.array_equal(vec["exc_off"], w_vec["exc_off"])
    if (want["scheme"] == 2).all():
        assert np.array_equal(packed, w_packed) and np.array_equal(exc, w_exc), "whole streams must be byte-identical"
    out = ctx.decode(dcol)
    ctx.synchronize()
    assert torch.equal(out.view(torch.int64), x.view(torch.int64))


@pytest.mark.parametrize("case", golden_io.rowgroup_samples(), ids=lambda c: c[0])
def test_golden_rowgroup_samples_encode_bit_exact(ctx, case):
    name, col_np, gold = case
    dcol, x = gpu_encode(ctx, col_np)
    got = layout.expand(*dcol.to_host())
    assert np.array_equal(got["k"], gold["k"]) and np.array_equal(got["combos"], gold["combos"])
    assert_parts_equal(got, gold, name)


def test_golden_first_vectors_encode(ctx):
    for name, col_np, gold, known in golden_io.first_vectors():
        dcol, x = gpu_encode(ctx, col_np)
        got = layout.expand(*dcol.to_host())
        assert got["scheme"][0] == gold["scheme"][0], name
        if gold["scheme"][0] == 2:
            assert np.array_equal(got["k"], gold["k"]) and np.array_equal(got["combos"], gold["combos"]), name
            assert_parts_equal(got, gold, name)
            if known[0] >= 0:  # the pairs test/test_alp_sample.cpp:178-179 asserts
                assert int(got["bw"][0]) == int(known[0]) and int(got["exc_cnt"][0]) == int(known[1]), name
        else:
            assert got["bw"][0] == gold["bw"][0] and got["lbw"][0] == gold["lbw"][0], name
            assert got["dict_size"][0] == gold["dict_size"][0] and np.array_equal(got["dict"][0], gold["dict"][0]), name
            assert np.array_equal(got["packed"], gold["packed"]) and np.array_equal(got["exc_cnt"], gold["exc_cnt"]), name
        out = ctx.decode(dcol)
        ctx.synchronize()
        assert torch.equal(out.view(torch.int64), x.view(torch.int64)), name


RD_COLUMNS = {
    "rd_unit": lambda: datagen.rd_column(130, seed=5, kind="unit"),
    "rd_latlon": lambda: datagen.rd_column(110, seed=6, kind="latlon"),
    "rd_few_left_parts": lambda: (np.random.default_rng(10).integers(0, 5, 150 * 1024).astype(np.float64) * 1e-3
                                  + np.random.default_rng(11).random(150 * 1024) * 1e-9),
    "mixed_alp_and_rd": lambda: np.concatenate([datagen.decimal_column(100, 2, seed=21), datagen.rd_column(100, seed=22),
                                                datagen.mixed_column(100, seed=23), datagen.rd_column(57, seed=24, kind="latlon")]),
}


@pytest.mark.parametrize("name", list(RD_COLUMNS.keys()))
def test_rd_vectors_with_reference_state_bit_exact(ctx, oracle, name):
    """ALP_RD vector encode given the rowgroup state the oracle (== reference) computed: right parts, exception
    lists and counts are bit-exact; left indices are compared at non-exception slots."""
    col_np = RD_COLUMNS[name]()
    want = oracle.encode_column(col_np)
    states, _, _, _ = layout.compact(want)
    dcol, x = gpu_encode(ctx, col_np, states=states)
    got = layout.expand(*dcol.to_host())
    assert_parts_equal(got, want, name)
    for v in np.nonzero(want["scheme"] == 1)[0]:
        a = oracle.unffor_u16(got["packed_left"][v], int(want["lbw"][v]))
        b = oracle.unffor_u16(want["packed_left"][v], int(want["lbw"][v]))
        keep = np.ones(1024, bool)
        keep[want["pos"][v, : int(want["exc_cnt"][v])]] = False
        assert np.array_equal(a[keep], b[keep]), f"{name}: left dictionary indices differ in vector {v}"
    out = ctx.decode(dcol)
    ctx.synchronize()
    assert torch.equal(out.view(torch.int64), x.view(torch.int64))


@pytest.mark.parametrize("name", list(RD_COLUMNS.keys()))
def test_rd_rowgroup_init_matches_reference_decisions(ctx, oracle, name):
    """Own rowgroup init: scheme, cut (right/left bit width), dictionary size AND the dictionary entries in order equal the
    oracle's (== the reference's), and with the rowgroup's sorted-order table every stream byte is the reference's."""
    col_np = RD_COLUMNS[name]()
    want = oracle.encode_column(col_np)
    dcol, x = gpu_encode(ctx, col_np)
    rg, vec, packed, exc = dcol.to_host()
    w_rg, _, _, _ = layout.compact(want)
    assert np.array_equal(rg["scheme"], w_rg["scheme"])
    rd = rg["scheme"] == 1
    assert np.array_equal(rg["rd_rbw"][rd], w_rg["rd_rbw"][rd]) and np.array_equal(rg["rd_lbw"][rd], w_rg["rd_lbw"][rd])
    assert np.array_equal(rg["rd_dict_size"][rd], w_rg["rd_dict_size"][rd])
    # the dictionary itself, entry by entry: the order of equally frequent left parts is libstdc++'s in the reference
    #  and is replayed on the device (alp_amd/csrc/rd_dictionary_order.hpp)
    assert np.array_equal(rg["rd_dict"][rd], w_rg["rd_dict"][rd]), name
    got = layout.expand(rg, vec, packed, exc)
    assert_parts_equal(got, want, name)
    # ... and the packed left streams in EVERY bit, exception slots included (alpgpu_column.d_rd_order), so whole streams match
    assert np.array_equal(got["packed_left"], want["packed_left"]), f"{name}: packed left streams differ"
    w_rg2, w_vec, w_packed, w_exc = layout.compact(want)
    assert np.array_equal(vec["packed_off"], w_vec["packed_off"]) and np.array_equal(vec["exc_off"], w_vec["exc_off"])
    assert np.array_equal(packed, w_packed) and np.array_equal(exc, w_exc), f"{name}: whole streams must be byte-identical"
    assert np.array_equal(rg["k"][~rd], w_rg["k"][~rd]) and np.array_equal(rg["combos"][~rd], w_rg["combos"][~rd])
    out = ctx.decode(dcol)
    ctx.synchronize()
    assert torch.equal(out.view(torch.int64), x.view(torch.int64))


@pytest.mark.parametrize("name", ["mixed_1pct", "drifting_k", "adversarial"])
def test_two_pass_encode_gives_the_same_column(ctx, oracle, name):
    """ALPGPU_OPT_ENCODE_TWO_PASS (analysis + scan + pack) and the default single-pass encode are byte-identical"""
    from alp_amd import capi
    col_np = COLUMNS[name]()
    d1, x = gpu_encode(ctx, col_np)
    try:
        ctx.set_option(capi.OPT_ENCODE_TWO_PASS, 1)
        d2, _ = gpu_encode(ctx, col_np)
    finally:
        ctx.set_option(capi.OPT_ENCODE_TWO_PASS, 0)
    for a, b in zip(d1.to_host(), d2.to_host()):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))


@pytest.mark.parametrize("name", ["mixed_1pct", "mixed_10pct", "drifting_k", "adversarial", "rd_unit", "every_width", "long_mixed"])
def test_classic_single_pass_kernel_gives_the_same_column(ctx, name):
    """ALPGPU_OPT_ENCODE_KERNEL = ALPGPU_ENCODE_KERNEL_CLASSIC (round 3's k_encode_fused: integers and packed units in registers, two tiles per CU) is a
    shipped option: the same bytes as the default lean kernel on ALP, ALP_RD, every packed width, heavy exception loads and a column of > 1000 look-back
    tiles, and the decode of its column is the input."""
    from alp_amd import capi
    col_np = {"rd_unit": lambda: datagen.rd_column(210, seed=12), "every_width": lambda: datagen.every_bit_width_column(),
              "long_mixed": lambda: datagen.mixed_column(9000, seed=78, exc_rate=0.03)}.get(name, COLUMNS.get(name))()
    d1, x = gpu_encode(ctx, col_np)
    try:
        ctx.set_option(capi.OPT_ENCODE_KERNEL, 1)
        d2, _ = gpu_encode(ctx, col_np)
        out = ctx.decode(d2)
        ctx.synchronize()
    finally:
        ctx.set_option(capi.OPT_ENCODE_KERNEL, 0)
    for a, b in zip(d1.to_host(), d2.to_host()):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
    assert torch.equal(out.view(torch.int64), x.view(torch.int64))


def test_long_column_spans_several_fused_launches_worth_of_tiles(ctx, oracle):
    """many tiles: the look-back chain across > 1000 workgroups must give the oracle's offsets"""
    col_np = datagen.mixed_column(6000, seed=77, exc_rate=0.02)
    want = oracle.encode_column(col_np)
    dcol, x = gpu_encode(ctx, col_np)
    rg, vec, packed, exc = dcol.to_host()
    w_rg, w_vec, w_packed, w_exc = layout.compact(want)
    assert np.array_equal(vec["packed_off"], w_vec["packed_off"]) and np.array_equal(vec["exc_off"], w_vec["exc_off"])
    assert np.array_equal(packed, w_packed) and np.array_equal(exc, w_exc)


def _streams(dcol):
    rg, vec, packed, exc = dcol.to_host()
    return [rg.view(np.uint8), vec.view(np.uint8), packed, exc]


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_large_column_every_encode_route_gives_the_same_bytes(ctx, dtype):
    """300k vectors (75k tiles, > 1000 look-back blocks): alpgpu_encode_*, rowgroup init + vector encode called separately and
    (double) the two-pass encode must produce byte-identical columns, twice in a row, and decode back to the input bits.
    Guards the ordered-offset protocol against races that small columns do not expose (one such race was found this way)."""
    from alp_amd import capi
    n = 300_000
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    if dtype == "f64":
        x = torch.round((torch.rand(n * 1024, dtype=torch.float64, device="cuda", generator=g) - 0.5) * 2e7) / 100.0
        m = torch.rand(n * 1024, device="cuda", generator=g) < 0.01
        x[m] = x[m] * 3.141592653589793
        it = torch.int64
    else:
        x = (torch.round(torch.rand(n * 1024, dtype=torch.float64, device="cuda", generator=g) * 1e5) / 100).to(torch.float32)
        it = torch.int32
    # ALP_RD rowgroups in the middle
    x[120_000 * 1024: 150_000 * 1024] = torch.rand(30_000 * 1024, dtype=x.dtype, device="cuda", generator=g)

    def encode_all():
        col = capi.DeviceColumn(n, dtype=dtype)
        ctx.encode(x, col)
        ctx.synchronize()
        assert ctx.column_totals(col)[2] == 0
        return col

    ref = encode_all()
    want = _streams(ref)
    out = ctx.decode(ref)
    ctx.synchronize()
    assert torch.equal(out.view(it), x.view(it))
    del out
    for a, b in zip(_streams(encode_all()), want):  # run to run
        assert np.array_equal(a, b)
    plain = capi.DeviceColumn(n, dtype=dtype)  # the two halves called separately
    ctx.rowgroup_init(x, plain)
    ctx.encode_vectors(x, plain)
    ctx.synchronize()
    for a, b in zip(_streams(plain), want):
        assert np.array_equal(a, b)
    if dtype == "f64":
        try:
            ctx.set_option(capi.OPT_ENCODE_TWO_PASS, 1)
            two = encode_all()
        finally:
            ctx.set_option(capi.OPT_ENCODE_TWO_PASS, 0)
        for a, b in zip(_streams(two), want):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_column_longer_than_one_fused_launch(ctx, dtype):
    """more than 2^20 vectors: the single-pass encode chains several launches through d_totals; offsets must continue across
    the seam, the round trip must be exact and the descriptors' extents must tile the streams without gaps"""
    from alp_amd import capi
    n = (1 << 20) + 4100
    g = torch.Generator(device="cuda")
    g.manual_seed(9)
    x = (torch.round(torch.rand(n * 1024, dtype=torch.float64, device="cuda", generator=g) * 1e5) / 100)
    x = x.to(torch.float32) if dtype == "f32" else x
    col = capi.DeviceColumn(n, dtype=dtype)
    ctx.encode(x, col)
    ctx.synchronize()
    pb, eb, ov = ctx.column_totals(col)
    assert ov == 0
    out = ctx.decode(col)
    ctx.synchronize()
    assert torch.equal(out.view(torch.int32 if dtype == "f32" else torch.int64), x.view(torch.int32 if dtype == "f32" else torch.int64))
    vec = col.vectors.cpu().numpy().view(capi.VECTOR_DTYPE)[:n]
    W = 4 if dtype == "f32" else 8
    psz, esz = layout.record_sizes(vec["scheme"], vec["bw"], vec["lbw"], vec["exc_cnt"], W)
    assert np.array_equal(vec["packed_off"], np.concatenate([[0], np.cumsum(psz)[:-1]]).astype(np.uint64))
    assert np.array_equal(vec["exc_off"], np.concatenate([[0], np.cumsum(esz)[:-1]]).astype(np.uint64))
    assert int(psz.sum()) == pb and int(esz.sum()) == eb


@pytest.mark.parametrize("n_vectors", [4, 5, 252, 255, 256, 257, 260, 511, 512, 513, 16383, 16384, 16385, 16388])
def test_look_back_block_boundaries(ctx, oracle, n_vectors):
    """column lengths around the look-back's block (64 tiles = 256 vectors) and tile (4 vectors) boundaries: offsets and
    whole streams against the oracle"""
    col_np = datagen.mixed_column(n_vectors, seed=1000 + n_vectors, exc_rate=0.02)
    want = oracle.encode_column(col_np)
    dcol, x = gpu_encode(ctx, col_np)
    rg, vec, packed, exc = dcol.to_host()
    w_rg, w_vec, w_packed, w_exc = layout.compact(want)
    assert np.array_equal(vec["packed_off"], w_vec["packed_off"]) and np.array_equal(vec["exc_off"], w_vec["exc_off"])
    assert np.array_equal(packed, w_packed) and np.array_equal(exc, w_exc)


def test_search_boundary_values_first_fitting_candidate(ctx, oracle):
    """The (e,f) search keeps its arithmetic in doubles wherever that is provably the reference's result and falls back to the
    literal int64 path on the two ambiguous boundaries (init_kernels.hip, PrecF64::step).  One constant sample set per boundary
    value: the state then names the first candidate that round-trips the value, or ALP_RD when none does."""
    from alp_amd import capi
    vals = datagen.search_boundary_values()
    vals = vals[np.unique(vals.view(np.uint64), return_index=True)[1]]
    states = torch.zeros((vals.size, 32), dtype=torch.uint8, device="cuda")
    smp = torch.from_numpy(np.repeat(vals, 32)).cuda()
    for i in range(vals.size):
        ctx.state_from_samples(smp[32 * i: 32 * i + 32], states[i])
    ctx.synchronize()
    got = states.cpu().numpy().reshape(-1).view(capi.ROWGROUP_DTYPE)
    for i, v in enumerate(vals):
        w_rg = layout.compact(oracle.encode_column(np.full(1024, v)))[0][0]
        assert got["scheme"][i] == w_rg["scheme"], f"value {v!r}: scheme"
        if w_rg["scheme"] == 2:
            assert got["k"][i] == w_rg["k"] and np.array_equal(got["combos"][i], w_rg["combos"]), f"value {v!r}: {got['combos'][i]} != {w_rg['combos']}"


@pytest.mark.parametrize("seed", list(range(60)))
def test_search_boundary_mixtures_whole_streams(ctx, oracle, seed):
    col_np = datagen.search_boundary_mixtures(seed)
    want = layout.compact(oracle.encode_column(col_np))
    dcol, x = gpu_encode(ctx, col_np)
    for a, b, what in zip(dcol.to_host(), want, ("rowgroup states", "descriptors", "packed stream", "exception stream")):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"seed {seed}: {what}"
    out = ctx.decode(dcol)
    ctx.synchronize()
    assert torch.equal(out.view(torch.int64), x.view(torch.int64))


@pytest.mark.parametrize("two_pass", [0, 1])
def test_exception_record_pad_bytes_are_zero_whatever_was_in_the_buffer(ctx, oracle, two_pass):
    """the exception stream is byte-reproducible even into a dirty buffer: both encode forms write the pad of every record"""
    from alp_amd import capi
    col_np = np.concatenate([datagen.mixed_column(130, seed=5, exc_rate=0.03), datagen.rd_column(110, seed=6)])
    want = layout.compact(oracle.encode_column(col_np))
    x = torch.from_numpy(col_np).cuda()
    dcol = capi.DeviceColumn(col_np.size // 1024)
    dcol.exc.fill_(0xA5)
    dcol.packed.fill_(0x5A)
    try:
        ctx.set_option(capi.OPT_ENCODE_TWO_PASS, two_pass)
        ctx.encode(x, dcol)
        ctx.synchronize()
    finally:
        ctx.set_option(capi.OPT_ENCODE_TWO_PASS, 0)
    for a, b, what in zip(dcol.to_host(), want, ("rowgroup states", "descriptors", "packed stream", "exception stream")):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"two_pass={two_pass}: {what}"


def _column_with_spoiled_samples(dtype, spoiled, n_vectors=212, seed=5):
    """Two-decimal values; `spoiled` = {sampled vector index: how many of its first-level samples (values 32*s of vector 12*sv) are
    replaced by full-mantissa noise, from sample 0 on} — in both rowgroups of the column."""
    rng = np.random.default_rng(seed)
    col = np.round(rng.uniform(-500.0, 500.0, n_vectors * 1024), 2).astype(dtype)
    noise = (rng.random(n_vectors * 1024) * 3.0 + 1.0 / 3.0).astype(dtype)
    for rg0 in range(0, n_vectors, 100):
        for sv, n_bad in spoiled.items():
            v = rg0 + 12 * sv
            if v < n_vectors:
                idx = v * 1024 + 32 * np.arange(n_bad)
                col[idx] = noise[idx]
    return col


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("spoiled", [{1: 32}, {0: 32, 3: 32}, {2: 20, 5: 24}, {1: 19, 4: 15, 6: 14}, {0: 32, 1: 32, 2: 32, 3: 32, 4: 32, 5: 32, 6: 32, 7: 28, 8: 20},
                                     {i: 32 for i in range(9)}], ids=lambda d: "-".join(f"{k}x{v}" for k, v in d.items()))
def test_rowgroups_whose_sampled_vectors_disagree(ctx, oracle, dtype, spoiled):
    """Rowgroups in which some sampled vectors are noise (every candidate of the (e,f) search far above the ALP_RD threshold) and the others
    decimals: the rowgroup is ALP all the same, and the noise vectors' votes — the cheapest of their hopeless candidates, ties by larger e
    then larger f — still count towards k and the candidate order.  Half-spoiled sampled vectors (19 / 20 / 24 noise samples of 32) sit on
    both sides of the threshold; all noise: ALP_RD.  (Written for an early-exit of the search that was measured and not kept,
    profiles/r03_encode_levers.txt; kept as a parity case of its own.)"""
    from oracle.pyoracle import OracleF32
    np_dtype = np.float64 if dtype == "f64" else np.float32
    col_np = _column_with_spoiled_samples(np_dtype, spoiled)
    orc = oracle if dtype == "f64" else OracleF32()
    want = layout.compact(orc.encode_column(col_np), 8 if dtype == "f64" else 4)
    from alp_amd import capi
    x = torch.from_numpy(col_np).cuda()
    dcol = capi.DeviceColumn(col_np.size // 1024, 0, dtype=dtype)
    ctx.encode(x, dcol)
    ctx.synchronize()
    for a, b, what in zip(dcol.to_host(), want, ("rowgroup states", "descriptors", "packed stream", "exception stream")):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"{dtype} {spoiled}: {what}"


@pytest.mark.parametrize("exceptions", [False, True], ids=["clean", "with_exceptions"])
@pytest.mark.parametrize("route", ["single_pass", "two_pass"])
def test_alp_vectors_of_every_packed_width(ctx, oracle, route, exceptions):
    """ffor (src/fastlanes_generated_ffor.cpp) at every width 0..64 THROUGH the column encoders — the single pass packs by scattering each
    lane's pairs into an LDS image (encode_device.hpp: pack_u64_scatter), the two-pass route by gathering per output word — on vectors
    whose width is chosen freely under an (e,f) = (0,0) rowgroup state the reference's search itself arrives at (datagen)."""
    from alp_amd import capi
    col_np = datagen.every_bit_width_column(exceptions=exceptions)
    want_o = oracle.encode_column(col_np)
    assert sorted(set(want_o["bw"].tolist())) == list(range(65)) and (want_o["scheme"] == 2).all()
    want = layout.compact(want_o)
    ctx.set_option(capi.OPT_ENCODE_TWO_PASS, 1 if route == "two_pass" else 0)
    try:
        dcol, x = gpu_encode(ctx, col_np)
    finally:
        ctx.set_option(capi.OPT_ENCODE_TWO_PASS, 0)
    for a, b, what in zip(dcol.to_host(), want, ("rowgroup states", "descriptors", "packed stream", "exception stream")):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"{route}: {what}"
    for vpw in (1, 2, 4):
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
        out = ctx.decode(dcol)
        ctx.synchronize()
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)
        assert torch.equal(out.view(torch.int64), x.view(torch.int64)), f"decode, {vpw} vectors per workgroup"


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_rd_dictionary_of_one_cut_matches_the_reference_formula(ctx, dtype):
    """alpgpu_rd_dictionary_for_cut_* = rd_encoder::build_left_parts_dictionary (rd.hpp:33-87) for ONE cut: its estimate against the reference's
    arithmetic restated in numpy (histogram of the left parts, the 8 most frequent stay, 32 bits per other sample; ties do not move the
    estimate), for every cut of 1..16 bits; and the first strictly smallest estimate is the cut alpgpu_rd_state_from_samples_* chooses."""
    from alp_amd import capi
    bits = 64 if dtype == "f64" else 32
    rng = np.random.default_rng(12)
    tdt, ndt, udt = (torch.float64, np.float64, np.uint64) if dtype == "f64" else (torch.float32, np.float32, np.uint32)
    for n_smp, gen in ((288, lambda: rng.random(288)), (288, lambda: rng.normal(0, 1e-3, 288) + 7.25), (96, lambda: rng.random(96) * 1e6), (17, lambda: rng.random(17))):
        smp = gen().astype(ndt)
        d_smp = torch.from_numpy(smp).cuda()
        st = torch.zeros(32, dtype=torch.uint8, device="cuda")
        est = torch.zeros(1, dtype=torch.float64, device="cuda")
        ests = []
        for cut in range(1, 17):
            rbw = bits - cut
            ctx.rd_dictionary_for_cut(d_smp, rbw, st, est)
            ctx.synchronize()
            left = smp.view(udt) >> udt(rbw)
            _, counts = np.unique(left, return_counts=True)
            counts = np.sort(counts)[::-1]
            ds = min(8, counts.size)
            lbw = max(1, int(np.ceil(np.log2(ds))))
            want = rbw + lbw + float(counts[8:].sum()) * 32 / n_smp
            got_state = st.cpu().numpy().view(capi.ROWGROUP_DTYPE)[0]
            assert float(est[0]) == want, (dtype, n_smp, cut, float(est[0]), want)
            assert (got_state["rd_rbw"], got_state["rd_lbw"], got_state["rd_dict_size"], got_state["scheme"]) == (rbw, lbw, ds, capi.SCHEME_ALP_RD)
            assert set(int(x) for x in got_state["rd_dict"][:ds]) <= set(int(x) for x in left)
            ests.append(float(est[0]))
        ctx.state_from_samples(d_smp, st, rd_only=True)
        ctx.synchronize()
        chosen = int(st.cpu().numpy().view(capi.ROWGROUP_DTYPE)[0]["rd_rbw"])
        assert chosen == bits - (1 + int(np.argmin(ests))), "find_best_dictionary takes the first strictly smaller estimate in cut order"
    with pytest.raises(capi.AlpGpuError, match="right_bit_width"):
        ctx.rd_dictionary_for_cut(d_smp, bits - 17, st, est)


def _assert_records_tile_the_streams(vec, pb, eb, value_bytes=8):
    """every vector's two records lie inside the streams, no two overlap, together they are the streams (no gap), and the eight vectors of a tile
    are adjacent and in order — what ALPGPU_OPT_ENCODE_UNORDERED promises about WHERE records lie"""
    psz, esz = layout.record_sizes(vec["scheme"], vec["bw"], vec["lbw"], vec["exc_cnt"], value_bytes)
    assert int(psz.sum()) == pb and int(esz.sum()) == eb
    for off, sz, total in ((vec["packed_off"].astype(np.int64), psz, pb), (vec["exc_off"].astype(np.int64), esz, eb)):
        nz = sz > 0
        o, s = off[nz], sz[nz]
        order = np.argsort(o, kind="stable")
        o, s = o[order], s[order]
        if o.size:
            assert o[0] == 0 and np.array_equal(o[1:], (o + s)[:-1]) and o[-1] + s[-1] == total
    n = vec.size
    for off, sz in ((vec["packed_off"].astype(np.int64), psz), (vec["exc_off"].astype(np.int64), esz)):
        for w in range(1, 8):  # vector 8t + w follows vector 8t + w - 1 (launches restart the tile count every 2^20 vectors: 2^20 is a multiple of 8)
            idx = np.arange(w, n, 8)
            assert np.array_equal(off[idx], off[idx - 1] + sz[idx - 1])


@pytest.mark.parametrize("name", ["mixed_1pct", "mixed_10pct", "drifting_k", "adversarial", "tiny", "rd_mix"])
def test_unordered_encode_writes_the_same_records_somewhere_else(ctx, oracle, name):
    """ALPGPU_OPT_ENCODE_UNORDERED: tiles reserve their bytes with one atomic add instead of the look-back.  Descriptor fields, packed words,
    exception values and positions of EVERY vector are the oracle's; the records tile the streams without gaps or overlaps; decode gives the input"""
    from alp_amd import capi
    col_np = np.concatenate([datagen.mixed_column(130, seed=31), datagen.rd_column(100, seed=32), datagen.drifting_column(70, seed=33)]) if name == "rd_mix" else COLUMNS[name]()
    want = oracle.encode_column(col_np)
    try:
        ctx.set_option(capi.OPT_ENCODE_UNORDERED, 1)
        dcol, x = gpu_encode(ctx, col_np)
    finally:
        ctx.set_option(capi.OPT_ENCODE_UNORDERED, 0)
    rg, vec, packed, exc = dcol.to_host()
    got = layout.expand(rg, vec, packed, exc)
    assert np.array_equal(got["k"], want["k"]) and np.array_equal(got["combos"], want["combos"])
    assert_parts_equal(got, want, name)
    assert np.array_equal(got["packed_left"], want["packed_left"])
    pb, eb, ov = ctx.column_totals(dcol)
    assert ov == 0
    _assert_records_tile_the_streams(vec, pb, eb)
    out = ctx.decode(dcol)
    ctx.synchronize()
    assert torch.equal(out.view(torch.int64), x.view(torch.int64))
    blob = ctx.to_blob(dcol, col_np.size)  # the container takes such a column as it is
    back, n_values = ctx.from_blob(blob)
    out2 = ctx.decode(back)
    ctx.synchronize()
    assert n_values == col_np.size and torch.equal(out2.view(torch.int64), x.view(torch.int64))
    # the chunked host route uploads a blob's streams in pieces and relies on offsets that ascend with the vector index: it must either refuse
    # such a blob with its own error or (a column whose records happen to be in order) decode it right — never read outside what it uploaded
    from alp_amd import capi as _capi
    host = torch.empty(col_np.size, dtype=torch.float64)
    try:
        nv = ctx.decompress_host(torch.from_numpy(blob), host)
        assert nv == col_np.size and np.array_equal(host.numpy().view(np.uint64), col_np.view(np.uint64))
    except _capi.AlpGpuError as exc:
        assert "chunk" in str(exc) or "ascend" in str(exc), str(exc)


def test_unordered_encode_of_a_large_column(ctx):
    """1.05 Mi vectors (two chained launches, the search beside the encode, ALP_RD rowgroups in the middle): per-vector records equal the ordered
    form's (compared through their offsets on the device), the streams are tiled, the round trip is exact; and the consumers read it"""
    from alp_amd import capi
    n = (1 << 20) + 4100
    g = torch.Generator(device="cuda")
    g.manual_seed(15)
    x = torch.round((torch.rand(n * 1024, dtype=torch.float64, device="cuda", generator=g) - 0.5) * 2e7) / 100.0
    m = torch.rand(n * 1024, device="cuda", generator=g) < 0.01
    x[m] = x[m] * 3.141592653589793
    del m
    x[500_000 * 1024: 520_000 * 1024] = torch.rand(20_000 * 1024, dtype=x.dtype, device="cuda", generator=g)
    ref = capi.DeviceColumn(n)
    ctx.encode(x, ref)
    ctx.synchronize()
    try:
        ctx.set_option(capi.OPT_ENCODE_UNORDERED, 1)
        col = capi.DeviceColumn(n)
        ctx.encode(x, col)
        ctx.synchronize()
    finally:
        ctx.set_option(capi.OPT_ENCODE_UNORDERED, 0)
    pb, eb, ov = ctx.column_totals(col)
    assert ov == 0 and (pb, eb) == ctx.column_totals(ref)[:2]
    vec = col.vectors.cpu().numpy().view(capi.VECTOR_DTYPE)[:n]
    rvec = ref.vectors.cpu().numpy().view(capi.VECTOR_DTYPE)[:n]
    for k in ("base", "bw", "e", "f", "lbw", "exc_cnt", "scheme"):
        assert np.array_equal(vec[k], rvec[k]), k
    assert np.array_equal(col.rowgroups.cpu().numpy(), ref.rowgroups.cpu().numpy())
    _assert_records_tile_the_streams(vec, pb, eb)
    assert not np.array_equal(vec["packed_off"], rvec["packed_off"]) or n < 64  # (it IS another order on any real run; not a requirement)
    # records byte for byte: a sample of vectors gathered from both columns by their own offsets
    psz, esz = layout.record_sizes(vec["scheme"], vec["bw"], vec["lbw"], vec["exc_cnt"])
    rng = np.random.default_rng(3)
    for v in np.concatenate([np.arange(0, 64), rng.integers(0, n, 2000), np.arange(n - 64, n), np.arange(500_000 - 8, 500_000 + 8), np.arange((1 << 20) - 16, (1 << 20) + 16)]):
        a, b, s = int(vec["packed_off"][v]), int(rvec["packed_off"][v]), int(psz[v])
        assert torch.equal(col.packed[a:a + s], ref.packed[b:b + s]), v
        a, b, s = int(vec["exc_off"][v]), int(rvec["exc_off"][v]), int(esz[v])
        assert torch.equal(col.exc[a:a + s], ref.exc[b:b + s]), v
    out = ctx.decode(col)
    ctx.synchronize()
    assert torch.equal(out.view(torch.int64), x.view(torch.int64))
    del out
    sums_a, sums_b = ctx.decode_sum(col), ctx.decode_sum(ref)
    ctx.synchronize()
    assert torch.equal(sums_a.view(torch.int64), sums_b.view(torch.int64))
