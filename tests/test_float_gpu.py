"""GPU parity of the single-precision path, through the C ABI, bit for bit:
decode (alpgpu_decode_f32) of oracle-encoded and golden columns; rowgroup init + encode (alpgpu_encode_f32) against the
oracle and the reference's golden outputs (incl. its own float test columns and asserted bit widths); batch primitives
for every 32-bit width; container round trip."""
import numpy as np
import pytest
import torch

import datagen
import golden_io
import layout

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def of32():
    from oracle.pyoracle import OracleF32
    return OracleF32()


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def gpu_decode(ctx, enc):
    from alp_amd import capi
    col = capi.DeviceColumn.from_host(*layout.compact(enc, 4), dtype="f32")
    out = ctx.decode(col)
    ctx.synchronize()
    return out.cpu().numpy()


def gpu_encode(ctx, col_np, states=None):
    from alp_amd import capi
    x = cu(col_np)
    col = capi.DeviceColumn(col_np.size // 1024, dtype="f32")
    if states is None:
        ctx.encode(x, col)
    else:
        col.rowgroups[: states.size * 32] = cu(states.view(np.uint8).reshape(-1))
        ctx.encode_vectors(x, col)
    ctx.synchronize()
    pb, eb, ov = ctx.column_totals(col)
    assert ov == 0
    return col, x


def assert_parts_equal(got, want, name):
    n = want["scheme"].size
    for k in ("scheme", "e", "f", "bw", "lbw", "base", "exc_cnt"):
        assert np.array_equal(got[k], want[k]), f"{name}: {k} differs at {np.nonzero(got[k] != want[k])[0][:5]}"
    assert np.array_equal(got["packed"], want["packed"]), f"{name}: packed words differ"
    for v in range(n):
        c = int(want["exc_cnt"][v])
        assert np.array_equal(got["pos"][v, :c], want["pos"][v, :c]), f"{name}: exception positions differ in vector {v}"
        w = np.uint32 if want["scheme"][v] == 2 else np.uint16
        assert np.array_equal(got["exc"][v].view(w)[:c], want["exc"][v].view(w)[:c]), f"{name}: exception values v{v}"


FLOATS = golden_io.float_vectors()


@pytest.mark.parametrize("case", FLOATS, ids=lambda c: c[0])
def test_golden_float_columns_decode_bit_exact(ctx, case):
    name, col, gold, _ = case
    got = gpu_decode(ctx, gold)
    assert np.array_equal(got.view(np.uint32), col.view(np.uint32)), name


@pytest.mark.parametrize("case", FLOATS, ids=lambda c: c[0])
def test_golden_float_columns_encode_bit_exact(ctx, case):
    """GPU rowgroup init + encode against the reference's outputs; ALP_RD rowgroups are compared in their decisions
    (cut, dictionary size, dictionary entries, right parts, exception lists) and round trip"""
    name, col, gold, known = case
    dcol, x = gpu_encode(ctx, col)
    got = layout.expand(*dcol.to_host(), 4)
    assert np.array_equal(got["scheme"], gold["scheme"]), name
    alp = gold["scheme"] == 2
    if alp.all():
        assert np.array_equal(got["k"], gold["k"]) and np.array_equal(got["combos"], gold["combos"]), name
        assert_parts_equal(got, gold, name)
    else:
        for k in ("bw", "lbw", "e", "f", "base"):
            assert np.array_equal(got[k], gold[k]), (name, k)
        assert np.array_equal(got["dict_size"], gold["dict_size"]) and np.array_equal(got["dict"], gold["dict"]), name
        assert np.array_equal(got["exc_cnt"], gold["exc_cnt"]) and np.array_equal(got["packed"], gold["packed"]), name
        assert np.array_equal(got["exc_cnt"][alp], gold["exc_cnt"][alp]) and np.array_equal(got["packed"][alp], gold["packed"][alp])
    if known[0] >= 0:
        assert int(got["bw"][0]) == int(known[0])
    if known[1] >= 0:
        assert int(got["exc_cnt"][0]) == int(known[1])
    out = ctx.decode(dcol)
    ctx.synchronize()
    assert torch.equal(out.view(torch.int32), x.view(torch.int32)), name


COLUMNS = {
    "decimal2": lambda: datagen.decimal_column_f32(230, 2, seed=1),
    "decimal1_wide": lambda: datagen.decimal_column_f32(120, 1, -1e5, 1e5, seed=12),
    "mixed_1pct": lambda: datagen.mixed_column_f32(250, seed=3, exc_rate=0.01),
    "mixed_30pct": lambda: datagen.mixed_column_f32(120, seed=4, exc_rate=0.30),
    "rd_unit": lambda: datagen.rd_column_f32(130, seed=5, kind="unit"),
    "rd_latlon": lambda: datagen.rd_column_f32(110, seed=6, kind="latlon"),
    "drifting_k": lambda: datagen.drifting_column_f32(200, seed=7),
    "integers": lambda: np.floor(datagen.decimal_column_f32(64, 0, 0, 1e6, seed=8)),
    "one_vector": lambda: datagen.decimal_column_f32(1, 3, seed=9),
    "adversarial": lambda: np.concatenate(list(datagen.adversarial_vectors_f32().values())),
    "negzero_samples": lambda: np.where(np.random.default_rng(13).random(150 * 1024) < 0.3, np.float32(-0.0),
                                        datagen.decimal_column_f32(150, 2, seed=14)).astype(np.float32),
}


@pytest.mark.parametrize("name", list(COLUMNS.keys()))
def test_synthetic_float_columns_decode_bit_exact(ctx, of32, name):
    col = COLUMNS[name]()
    enc = of32.encode_column(col)
    got = gpu_decode(ctx, enc)
    assert np.array_equal(got.view(np.uint32), of32.decode_column(enc).view(np.uint32))
    assert np.array_equal(got.view(np.uint32), col.view(np.uint32))


@pytest.mark.parametrize("vpw", [1, 2, 4])
def test_decode_launch_shapes_agree(ctx, of32, vpw):
    from alp_amd import capi
    col = np.concatenate([datagen.mixed_column_f32(103, seed=21, exc_rate=0.05), datagen.rd_column_f32(57, seed=22)])
    enc = of32.encode_column(col)
    try:
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
        got = gpu_decode(ctx, enc)
    finally:
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)
    assert np.array_equal(got.view(np.uint32), col.view(np.uint32))


@pytest.mark.parametrize("shape", [16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28])
def test_streamed_decode_bit_exact(ctx, of32, shape):
    """the persistent, streaming float decode (decode_stream_f32_kernels.hip; ALPGPU_OPT_DECODE_VECTORS_PER_WG 16-18: chunks of 8 / 16 / 4 vectors): every synthetic
    column, the golden float columns and vectors of every exception count, among them records that do not fit the chunk's arena (decoded from HBM directly)"""
    from alp_amd import capi
    try:
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, shape)
        for name, make in COLUMNS.items():
            col = make()
            got = gpu_decode(ctx, of32.encode_column(col))
            assert np.array_equal(got.view(np.uint32), col.view(np.uint32)), name
        for name, col, gold, _ in FLOATS:
            assert np.array_equal(gpu_decode(ctx, gold).view(np.uint32), col.view(np.uint32)), name
        # exception counts 0 .. 1024 in one column (positions spread over the vector), narrow and wide digits
        rng = np.random.default_rng(77)
        for decimals, hi in ((1, 10.0), (3, 1.0e4)):
            vecs = []
            for cnt in list(range(0, 70)) + [100, 127, 128, 129, 255, 256, 257, 511, 512, 700, 1023, 1024]:
                v = np.round(rng.uniform(0, hi, 1024), decimals).astype(np.float32)
                at = np.sort(rng.choice(1024, cnt, replace=False))
                v[at] = rng.standard_normal(cnt).astype(np.float32) * np.float32(1.2345678e-3)
                vecs.append(v)
            col = np.concatenate(vecs)
            enc = of32.encode_column(col)
            got = gpu_decode(ctx, enc)
            assert np.array_equal(got.view(np.uint32), col.view(np.uint32)), (decimals, np.nonzero(got.view(np.uint32) != col.view(np.uint32))[0][:8])
        # a longer column (more chunks than workgroups' first round), ALP and ALP_RD rowgroups alternating
        col = np.concatenate([datagen.mixed_column_f32(700, seed=31, exc_rate=0.02), datagen.rd_column_f32(300, seed=32), datagen.decimal_column_f32(1003, 1, seed=33)])
        x = cu(col)
        dcol = ctx.encode(x)
        out = ctx.decode(dcol)
        ctx.synchronize()
        assert torch.equal(out.view(torch.int32), x.view(torch.int32))
        # ... and the same column with its records NOT in vector order (ALPGPU_OPT_ENCODE_UNORDERED): a chunk's records are then no single span of the streams
        ctx.set_option(capi.OPT_ENCODE_UNORDERED, 1)
        try:
            dcol = ctx.encode(x)
        finally:
            ctx.set_option(capi.OPT_ENCODE_UNORDERED, 0)
        out = ctx.decode(dcol)
        ctx.synchronize()
        assert torch.equal(out.view(torch.int32), x.view(torch.int32))
    finally:
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)


@pytest.mark.parametrize("name", list(COLUMNS.keys()))
def test_synthetic_float_columns_encode_bit_exact(ctx, of32, name):
    col_np = COLUMNS[name]()
    want = of32.encode_column(col_np)
    dcol, x = gpu_encode(ctx, col_np)
    rg, vec, packed, exc = dcol.to_host()
    got = layout.expand(rg, vec, packed, exc, 4)
    w_rg, w_vec, w_packed, w_exc = layout.compact(want, 4)
    assert np.array_equal(rg["scheme"], w_rg["scheme"]), name
    alp_rg = rg["scheme"] == 2
    assert np.array_equal(rg["k"][alp_rg], w_rg["k"][alp_rg]) and np.array_equal(rg["combos"][alp_rg], w_rg["combos"][alp_rg]), name
    assert np.array_equal(rg["rd_rbw"], w_rg["rd_rbw"]) and np.array_equal(rg["rd_lbw"], w_rg["rd_lbw"]) and np.array_equal(rg["rd_dict_size"], w_rg["rd_dict_size"])
    assert np.array_equal(rg["rd_dict"], w_rg["rd_dict"]), "ALP_RD dictionaries must come out in the reference's order"
    assert_parts_equal(got, want, name)
    # ALP and ALP_RD alike: offsets and whole streams are the oracle's bytes (exception-slot left indices included)
    assert np.array_equal(vec["packed_off"], w_vec["packed_off"]) and np.array_equal(vec["exc_off"], w_vec["exc_off"])
    assert np.array_equal(packed, w_packed) and np.array_equal(exc, w_exc), "whole streams must be byte-identical"
    out = ctx.decode(dcol)
    ctx.synchronize()
    assert torch.equal(out.view(torch.int32), x.view(torch.int32))


@pytest.mark.parametrize("name", ["rd_unit", "rd_latlon", "mixed_1pct", "drifting_k"])
def test_float_vectors_with_reference_state_bit_exact(ctx, of32, name):
    """vector encode given the rowgroup states the oracle (== reference) computed, ALP_RD dictionaries included: every
    stream byte is the reference's except the left index at exception slots"""
    col_np = COLUMNS[name]()
    want = of32.encode_column(col_np)
    states, _, _, _ = layout.compact(want, 4)
    dcol, x = gpu_encode(ctx, col_np, states=states)
    got = layout.expand(*dcol.to_host(), 4)
    assert_parts_equal(got, want, name)
    from oracle.pyoracle import Oracle
    o16 = Oracle()
    for v in np.nonzero(want["scheme"] == 1)[0]:
        a = o16.unffor_u16(got["packed_left"][v], int(want["lbw"][v]))
        b = o16.unffor_u16(want["packed_left"][v], int(want["lbw"][v]))
        keep = np.ones(1024, bool)
        keep[want["pos"][v, : int(want["exc_cnt"][v])]] = False
        assert np.array_equal(a[keep], b[keep]), f"{name}: left dictionary indices differ in vector {v}"
    out = ctx.decode(dcol)
    ctx.synchronize()
    assert torch.equal(out.view(torch.int32), x.view(torch.int32))


def test_ffor_unffor_falp_u32_every_bit_width(ctx, of32):
    rng = np.random.default_rng(5)
    bws = np.arange(0, 33, dtype=np.uint8)
    n = bws.size
    base = rng.integers(-2**30, 2**30, n).astype(np.int32)
    vals = np.zeros((n, 1024), np.uint32)
    for i, bw in enumerate(bws):
        span = (1 << int(bw)) - 1
        vals[i] = (rng.integers(0, 2**32, 1024, dtype=np.uint64) & np.uint64(span)).astype(np.uint32) + np.uint32(int(base[i]) & 0xFFFFFFFF)
    want_packed = np.stack([of32.ffor_u32(vals[i], int(bws[i]), int(base[i])) for i in range(n)])
    d_packed = torch.zeros((n, 1024), dtype=torch.int32, device="cuda")
    ctx.ffor_i32(cu(vals.view(np.int32)), d_packed, cu(bws), cu(base))
    ctx.synchronize()
    got = d_packed.cpu().numpy().view(np.uint32)
    for i, bw in enumerate(bws):
        assert np.array_equal(got[i, :32 * int(bw)], want_packed[i, :32 * int(bw)]), f"ffor bw={bw}"
        assert not got[i, 32 * int(bw):].any(), f"ffor bw={bw} wrote past 32*bw words"
    d_out = torch.zeros((n, 1024), dtype=torch.int32, device="cuda")
    ctx.unffor_i32(cu(want_packed.view(np.int32)), d_out, cu(bws), cu(base))
    ctx.synchronize()
    assert np.array_equal(d_out.cpu().numpy().view(np.uint32), vals), "unffor"
    fac = rng.integers(0, 10, n).astype(np.uint8)
    exp = np.maximum(fac, rng.integers(0, 11, n)).astype(np.uint8)
    d_f = torch.zeros((n, 1024), dtype=torch.float32, device="cuda")
    ctx.falp_f32(cu(want_packed.view(np.int32)), d_f, cu(bws), cu(base), cu(fac), cu(exp))
    ctx.synchronize()
    want = np.stack([of32.falp(want_packed[i], int(bws[i]), int(base[i]), int(fac[i]), int(exp[i])) for i in range(n)])
    assert np.array_equal(d_f.cpu().numpy().view(np.uint32), want.view(np.uint32)), "falp"
    d_f2 = torch.zeros((n, 1024), dtype=torch.float32, device="cuda")
    ctx.decode_values_f32(cu(vals.view(np.int32)), d_f2, cu(fac), cu(exp))
    ctx.synchronize()
    assert np.array_equal(d_f2.cpu().numpy().view(np.uint32), want.view(np.uint32)), "decode"


def test_encode_simdized_analyze_patch_f32(ctx, of32):
    cases = datagen.adversarial_vectors_f32()
    efs = [(2, 0), (10, 10), (0, 0), (5, 2), (10, 0), (9, 9), (7, 4)]
    vecs, fac, exp = [], [], []
    for name, v in cases.items():
        for e, f in efs:
            vecs.append(v), fac.append(f), exp.append(e)
    x = np.stack(vecs)
    n = x.shape[0]
    d_exc = torch.zeros((n, 1024), dtype=torch.float32, device="cuda")
    d_pos = torch.zeros((n, 1024), dtype=torch.int16, device="cuda")
    d_cnt = torch.zeros(n, dtype=torch.int16, device="cuda")
    d_enc = torch.zeros((n, 1024), dtype=torch.int32, device="cuda")
    d_fac, d_exp = cu(np.array(fac, np.uint8)), cu(np.array(exp, np.uint8))
    ctx.encode_simdized_f32(cu(x), d_exc, d_pos, d_cnt, d_enc, d_fac, d_exp)
    d_bw = torch.zeros(n, dtype=torch.uint8, device="cuda")
    d_base = torch.zeros(n, dtype=torch.int32, device="cuda")
    ctx.analyze_ffor_i32(d_enc, d_bw, d_base)
    d_dec = torch.zeros((n, 1024), dtype=torch.float32, device="cuda")
    ctx.decode_values_f32(d_enc, d_dec, d_fac, d_exp)
    ctx.patch_f32(d_dec, d_exc, d_pos, d_cnt)
    ctx.synchronize()
    enc, exc, pos = d_enc.cpu().numpy(), d_exc.cpu().numpy(), d_pos.cpu().numpy().view(np.uint16)
    cnt = d_cnt.cpu().numpy().view(np.uint16)
    for i in range(n):
        we, wx, wp, wc = of32.encode_simdized(x[i], fac[i], exp[i])
        assert cnt[i] == wc and np.array_equal(enc[i], we), (i, fac[i], exp[i])
        assert np.array_equal(pos[i, :wc], wp[:wc]) and np.array_equal(exc[i, :wc].view(np.uint32), wx[:wc].view(np.uint32)), i
        assert (int(d_bw[i]), int(d_base[i])) == of32.analyze_ffor(we), i
    assert np.array_equal(d_dec.cpu().numpy().view(np.uint32), x.view(np.uint32)), "decode+patch must reproduce the input bits"


def test_encode_values_and_rd_vectors_f32(ctx, of32):
    col = datagen.drifting_column_f32(200, seed=7)
    want = of32.encode_column(col)
    states, _, _, _ = layout.compact(want, 4)
    n = 200
    alp = want["scheme"] == 2
    d_exc = torch.zeros((n, 1024), dtype=torch.float32, device="cuda")
    d_pos = torch.zeros((n, 1024), dtype=torch.int16, device="cuda")
    d_cnt = torch.zeros(n, dtype=torch.int16, device="cuda")
    d_enc = torch.zeros((n, 1024), dtype=torch.int32, device="cuda")
    d_fac = torch.zeros(n, dtype=torch.uint8, device="cuda")
    d_exp = torch.zeros(n, dtype=torch.uint8, device="cuda")
    if alp.any():
        sel = np.nonzero(alp)[0]
        idx = (sel // 100).astype(np.uint32)
        m = sel.size
        ctx.encode_values_f32(cu(col.reshape(n, 1024)[sel]), cu(states.view(np.uint8)), cu(idx.view(np.int32)), d_exc[:m], d_pos[:m], d_cnt[:m],
                              d_enc[:m], d_fac[:m], d_exp[:m])
        ctx.synchronize()
        assert np.array_equal(d_fac[:m].cpu().numpy(), want["f"][sel]) and np.array_equal(d_exp[:m].cpu().numpy(), want["e"][sel])
        assert np.array_equal(d_cnt[:m].cpu().numpy().view(np.uint16), want["exc_cnt"][sel])
    # ALP_RD vectors on unpacked arrays
    col = datagen.rd_column_f32(120, seed=5)
    want = of32.encode_column(col)
    states, _, _, _ = layout.compact(want, 4)
    n = 120
    d_exc = torch.zeros((n, 1024), dtype=torch.int16, device="cuda")
    d_pos = torch.zeros((n, 1024), dtype=torch.int16, device="cuda")
    d_cnt = torch.zeros(n, dtype=torch.int16, device="cuda")
    d_right = torch.zeros((n, 1024), dtype=torch.int32, device="cuda")
    d_left = torch.zeros((n, 1024), dtype=torch.int16, device="cuda")
    x = cu(col.reshape(n, 1024))
    st = cu(states.view(np.uint8))
    ctx.rd_encode_vectors_f32(x, st, None, d_exc, d_pos, d_cnt, d_right, d_left)
    d_out = torch.zeros((n, 1024), dtype=torch.float32, device="cuda")
    ctx.rd_decode_vectors_f32(d_out, d_right, d_left, st, None, d_exc, d_pos, d_cnt)
    ctx.synchronize()
    assert np.array_equal(d_cnt.cpu().numpy().view(np.uint16), want["exc_cnt"])
    right = d_right.cpu().numpy().view(np.uint32)
    for v in range(n):
        rbw = int(want["bw"][v])
        assert np.array_equal(right[v], col.view(np.uint32)[v * 1024:(v + 1) * 1024] & np.uint32((1 << rbw) - 1))
        c = int(want["exc_cnt"][v])
        assert np.array_equal(d_pos[v, :c].cpu().numpy().view(np.uint16), want["pos"][v, :c])
        assert np.array_equal(d_exc[v, :c].cpu().numpy().view(np.uint16), want["exc"][v].view(np.uint16)[:c])
    assert torch.equal(d_out.view(torch.int32), x.view(torch.int32))


def test_state_from_samples_f32(ctx, of32):
    """find_top_k_combinations / find_best_dictionary on caller-gathered samples (the per-rowgroup entry the header uses)"""
    for name in ("decimal2", "rd_unit", "drifting_k"):
        col = COLUMNS[name]()[: 100 * 1024]
        want = of32.encode_column(col)
        w_rg, _, _, _ = layout.compact(want, 4)
        smp = np.concatenate([col[(12 * w) * 1024:(12 * w + 1) * 1024:32] for w in range(9)]).astype(np.float32)
        st = torch.zeros(32, dtype=torch.uint8, device="cuda")
        ctx.state_from_samples(cu(smp), st)
        ctx.synchronize()
        from alp_amd import capi
        got = st.cpu().numpy().view(capi.ROWGROUP_DTYPE)[0]
        assert got["scheme"] == w_rg["scheme"][0], name
        if got["scheme"] == 2:
            assert got["k"] == w_rg["k"][0] and np.array_equal(got["combos"], w_rg["combos"][0]), name
        else:
            assert got["rd_rbw"] == w_rg["rd_rbw"][0] and got["rd_lbw"] == w_rg["rd_lbw"][0] and got["rd_dict_size"] == w_rg["rd_dict_size"][0], name
            assert np.array_equal(got["rd_dict"], w_rg["rd_dict"][0]), name


def test_float_container_round_trip_and_tail(ctx):
    from alp_amd import capi
    n_values = 5 * 1024 + 77
    col_np = datagen.mixed_column_f32(6, seed=31, exc_rate=0.02)[: 6 * 1024].copy()
    x = cu(col_np)
    ctx.pad_tail(x, n_values)
    ctx.synchronize()
    padded = x.cpu().numpy()
    assert np.array_equal(padded[:n_values].view(np.uint32), col_np[:n_values].view(np.uint32))
    assert (padded[n_values:].view(np.uint32) == padded[5 * 1024:5 * 1024 + 1].view(np.uint32)).all()
    dcol = ctx.encode(x)
    ctx.synchronize()
    blob = ctx.to_blob(dcol, n_values)
    col2, nv = ctx.from_blob(blob)
    assert nv == n_values and col2.dtype == "f32"
    out = ctx.decode(col2)
    ctx.synchronize()
    assert torch.equal(out.view(torch.int32), x.view(torch.int32))
    # a float blob is refused by the double entry point and vice versa
    with pytest.raises(capi.AlpGpuError):
        bad = capi.DeviceColumn(6)
        nvv = capi._u64()
        capi._check(capi.lib.alpgpu_column_from_blob(ctx.h, blob.ctypes.data_as(capi._vp), blob.size, capi.C.byref(bad.c), capi.C.byref(nvv)), "from_blob")


def test_long_float_column_lookback_offsets(ctx, of32):
    col_np = datagen.mixed_column_f32(3000, seed=77, exc_rate=0.02, decimals_per_rowgroup=(1, 2))
    want = of32.encode_column(col_np)
    # (the two-decimal rowgroups of this float column resolve to ALP_RD: both schemes alternate along the 3000 vectors, which is
    # what the ordered offsets have to get right)
    assert (want["scheme"] == 2).sum() >= 1000 and (want["scheme"] == 1).sum() >= 1000
    dcol, x = gpu_encode(ctx, col_np)
    rg, vec, packed, exc = dcol.to_host()
    w_rg, w_vec, w_packed, w_exc = layout.compact(want, 4)
    assert np.array_equal(vec["packed_off"], w_vec["packed_off"]) and np.array_equal(vec["exc_off"], w_vec["exc_off"])
    assert np.array_equal(packed, w_packed) and np.array_equal(exc, w_exc)


def test_float_exception_record_pad_bytes_are_zero_in_a_dirty_buffer(ctx, of32):
    from alp_amd import capi
    col_np = np.concatenate([datagen.mixed_column_f32(130, seed=5, exc_rate=0.03), datagen.rd_column_f32(110, seed=6)])
    want = layout.compact(of32.encode_column(col_np), 4)
    x = torch.from_numpy(col_np).cuda()
    dcol = capi.DeviceColumn(col_np.size // 1024, dtype="f32")
    dcol.exc.fill_(0xA5)
    dcol.packed.fill_(0x5A)
    ctx.encode(x, dcol)
    ctx.synchronize()
    for a, b, what in zip(dcol.to_host(), want, ("rowgroup states", "descriptors", "packed stream", "exception stream")):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), what


@pytest.mark.parametrize("name", ["mixed_1pct", "mixed_30pct", "rd_latlon", "drifting_k", "adversarial", "one_vector"])
def test_unordered_float_encode_writes_the_same_records_somewhere_else(ctx, of32, name):
    """ALPGPU_OPT_ENCODE_UNORDERED on a float column: every vector's descriptor fields, packed words and exception record are the oracle's; the records
    tile the two streams without gaps or overlaps, a tile's eight vectors adjacent and in order; the decode gives the input back"""
    from alp_amd import capi
    import test_encode_gpu
    col_np = COLUMNS[name]()
    want = of32.encode_column(col_np)
    try:
        ctx.set_option(capi.OPT_ENCODE_UNORDERED, 1)
        dcol, x = gpu_encode(ctx, col_np)
    finally:
        ctx.set_option(capi.OPT_ENCODE_UNORDERED, 0)
    rg, vec, packed, exc = dcol.to_host()
    got = layout.expand(rg, vec, packed, exc, 4)
    assert_parts_equal(got, want, name)
    assert np.array_equal(got["packed_left"], want["packed_left"])
    pb, eb, ov = ctx.column_totals(dcol)
    test_encode_gpu._assert_records_tile_the_streams(vec, pb, eb, value_bytes=4)
    out = ctx.decode(dcol)
    ctx.synchronize()
    assert torch.equal(out.view(torch.int32), x.view(torch.int32))
