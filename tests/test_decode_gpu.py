"""GPU parity: alpgpu_decode_f64 (fused falp + patch / RD glue) against the oracle and the golden vectors.
Inputs are encoded on the CPU by the oracle (pinned against the reference), uploaded in the compact HBM
layout, decoded by the HIP kernel through the C ABI, and compared bit for bit."""
import numpy as np
import pytest

import datagen
import golden_io
import layout

pytestmark = pytest.mark.gpu


def gpu_decode(ctx, enc):
    from alp_amd import capi
    rg, vec, packed, exc = layout.compact(enc)
    col = capi.DeviceColumn.from_host(rg, vec, packed, exc)
    out = ctx.decode(col)
    ctx.synchronize()
    return out.cpu().numpy()


def test_golden_first_vectors_decode_bit_exact(ctx):
    cases = golden_io.first_vectors()
    # all 98 reference columns as ONE device column is not possible (state is per rowgroup): decode one by one
    for name, col, gold, _ in cases:
        got = gpu_decode(ctx, gold)
        assert np.array_equal(got.view(np.uint64), col.view(np.uint64)), name


@pytest.mark.parametrize("case", golden_io.rowgroup_samples(), ids=lambda c: c[0])
def test_golden_rowgroup_samples_decode_bit_exact(ctx, case):
    name, col, gold = case
    got = gpu_decode(ctx, gold)
    assert np.array_equal(got.view(np.uint64), col.view(np.uint64)), name


COLUMNS = {
    "decimal2": lambda: datagen.decimal_column(230, 2, seed=1),
    "mixed_1pct": lambda: datagen.mixed_column(250, seed=3, exc_rate=0.01),
    "mixed_30pct": lambda: datagen.mixed_column(120, seed=4, exc_rate=0.30),  # > 128 exceptions/vector: HBM fetch path
    "rd_unit": lambda: datagen.rd_column(130, seed=5, kind="unit"),
    "rd_latlon": lambda: datagen.rd_column(110, seed=6, kind="latlon"),
    "drifting_k": lambda: datagen.drifting_column(200, seed=7),
    "one_vector": lambda: datagen.decimal_column(1, 3, seed=9),
    "adversarial": lambda: np.concatenate(list(datagen.adversarial_vectors().values())),
}


@pytest.mark.parametrize("name", list(COLUMNS.keys()))
def test_synthetic_columns_decode_bit_exact(ctx, oracle, name):
    col = COLUMNS[name]()
    enc = oracle.encode_column(col)
    want = oracle.decode_column(enc)
    got = gpu_decode(ctx, enc)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    assert np.array_equal(got.view(np.uint64), col.view(np.uint64))


@pytest.mark.parametrize("bw", [0, 1, 2, 3, 7, 8, 13, 16, 17, 31, 32, 33, 47, 52, 53, 60, 63, 64])
def test_falp_every_bit_width_class(ctx, oracle, bw):
    """hand-built ALP vectors at a chosen bit width (random packed words are a valid FFOR stream), with
    wrap-around base and exceptions at random positions; oracle falp+patch is the expectation."""
    rng = np.random.default_rng(1000 + bw)
    n = 37
    enc = dict(scheme=np.full(n, 2, np.uint8), e=np.zeros(n, np.uint8), f=np.zeros(n, np.uint8), bw=np.full(n, bw, np.uint8),
               lbw=np.zeros(n, np.uint8), base=rng.integers(-2**62, 2**62, n), exc_cnt=np.zeros(n, np.uint16),
               packed=np.zeros((n, 1024), np.int64), packed_left=np.zeros((n, 1024), np.uint16),
               exc=np.zeros((n, 1024), np.float64), pos=np.zeros((n, 1024), np.uint16), dict=np.zeros((1, 8), np.uint16),
               dict_size=np.zeros(1, np.uint8), k=np.ones(1, np.uint8), combos=np.zeros((1, 10), np.int32))
    for v in range(n):
        e = int(rng.integers(0, 19)); f = int(rng.integers(0, e + 1))
        enc["e"][v], enc["f"][v] = e, f
        words = rng.integers(-2**63, 2**63 - 1, 16 * bw, dtype=np.int64)
        enc["packed"][v, :16 * bw] = words
        c = int(rng.choice([0, 1, 5, 64, 129, 1024]))
        pos = np.sort(rng.choice(1024, c, replace=False)).astype(np.uint16)
        enc["exc_cnt"][v] = c
        enc["pos"][v, :c] = pos
        enc["exc"][v, :c] = rng.integers(0, 2**64, c, dtype=np.uint64).view(np.float64)  # arbitrary bit patterns incl. NaN payloads
    want = oracle.decode_column(enc)
    got = gpu_decode(ctx, enc)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    from alp_amd import capi
    try:  # four vectors per workgroup over the narrow stage (widths <= 17 staged, wider ones read straight from HBM): the same bits
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 4)
        got4 = gpu_decode(ctx, enc)
    finally:
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)
    assert np.array_equal(got4.view(np.uint64), want.view(np.uint64))


@pytest.mark.parametrize("vectors_per_wg,plain", [(2, 0), (2, 1), (1, 1), (4, 0), (4, 1)])  # 4: the narrow stage; wide and ALP_RD vectors take its straight-from-HBM arm
def test_tuning_options_do_not_change_results(ctx, oracle, vectors_per_wg, plain):
    from alp_amd import capi
    try:
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vectors_per_wg)
        ctx.set_option(capi.OPT_DECODE_PLAIN_STORES, plain)
        for name in ("mixed_1pct", "mixed_30pct", "rd_latlon", "adversarial", "one_vector"):
            col = COLUMNS[name]()
            enc = oracle.encode_column(col)
            got = gpu_decode(ctx, enc)
            assert np.array_equal(got.view(np.uint64), col.view(np.uint64)), (name, vectors_per_wg, plain)
    finally:
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)
        ctx.set_option(capi.OPT_DECODE_PLAIN_STORES, 0)


@pytest.mark.parametrize("pad_kib", [0, 3, 6, 11, 14, 40, 120])
def test_residency_pad_does_not_change_results(ctx, oracle, pad_kib):
    """ALPGPU_OPT_DECODE_RESIDENCY_PAD (unused dynamic LDS per store-decode workgroup = fewer workgroups resident per CU; the rule's own choices are 0 / 3 / 6 /
    11 / 14 KiB): never changes a byte, at one and at two vectors per workgroup, up to a pad that leaves ONE workgroup per CU."""
    from alp_amd import capi
    try:
        ctx.set_option(capi.OPT_DECODE_RESIDENCY_PAD, pad_kib)
        for vpw in (1, 2):
            ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
            for name in ("mixed_1pct", "mixed_30pct", "rd_latlon", "adversarial"):
                col = COLUMNS[name]()
                got = gpu_decode(ctx, oracle.encode_column(col))
                assert np.array_equal(got.view(np.uint64), col.view(np.uint64)), (name, pad_kib, vpw)
    finally:
        ctx.set_option(capi.OPT_DECODE_RESIDENCY_PAD, -1)
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)


@pytest.mark.parametrize("pairing", [1, 2, 3])
def test_pairing_workgroups_write_the_same_bytes(ctx, oracle, pairing):
    """ALPGPU_OPT_DECODE_PAIRING (k_decode_pairs: workgroups that own two vectors and decide from their descriptors how to run them): every
    column class incl. odd vector counts, ALP_RD, exception-heavy and adversarial vectors decodes to the same bits as the default shape"""
    from alp_amd import capi
    try:
        ctx.set_option(capi.OPT_DECODE_PAIRING, pairing)
        for name in ("mixed_1pct", "mixed_30pct", "rd_latlon", "adversarial", "one_vector", "decimal2"):
            if name not in COLUMNS:
                continue
            col = COLUMNS[name]()
            for cut in (len(col), max(1024, (len(col) // 1024 - 1) * 1024), 3 * 1024, 4 * 1024, 5 * 1024):  # even, odd and short vector counts
                part = col[: min(cut, len(col))]
                if len(part) == 0:
                    continue
                enc = oracle.encode_column(part)
                got = gpu_decode(ctx, enc)
                assert np.array_equal(got.view(np.uint64), part.view(np.uint64)), (name, pairing, len(part))
    finally:
        ctx.set_option(capi.OPT_DECODE_PAIRING, 0)


def _alp_vectors_with_exception_counts(rng, counts, bw, placement="random"):
    """hand-built ALP vectors (random packed words) with the given exception counts; placement: random / front (all in the first quarter) /
    edges (quarter boundaries first: 0, 255, 256, 511, 512, 767, 768, 1023)"""
    n = len(counts)
    enc = dict(scheme=np.full(n, 2, np.uint8), e=np.zeros(n, np.uint8), f=np.zeros(n, np.uint8), bw=np.full(n, bw, np.uint8),
               lbw=np.zeros(n, np.uint8), base=rng.integers(-2**40, 2**40, n), exc_cnt=np.zeros(n, np.uint16),
               packed=np.zeros((n, 1024), np.int64), packed_left=np.zeros((n, 1024), np.uint16),
               exc=np.zeros((n, 1024), np.float64), pos=np.zeros((n, 1024), np.uint16), dict=np.zeros((1, 8), np.uint16),
               dict_size=np.zeros(1, np.uint8), k=np.ones(1, np.uint8), combos=np.zeros((1, 10), np.int32))
    edges = np.array([0, 255, 256, 511, 512, 767, 768, 1023])
    for v, c in enumerate(counts):
        e = int(rng.integers(0, 19)); f = int(rng.integers(0, e + 1))
        enc["e"][v], enc["f"][v] = e, f
        enc["packed"][v, :16 * bw] = rng.integers(-2**63, 2**63 - 1, 16 * bw, dtype=np.int64)
        if placement == "front":
            pos = np.sort(rng.choice(256, min(c, 256), replace=False))
            c = len(pos)
        elif placement == "edges":
            rest = np.setdiff1d(np.arange(1024), edges)
            pos = np.sort(np.concatenate([edges[:min(c, 8)], rng.choice(rest, max(c - 8, 0), replace=False)]))
        else:
            pos = np.sort(rng.choice(1024, c, replace=False))
        enc["exc_cnt"][v] = c
        enc["pos"][v, :c] = pos.astype(np.uint16)
        enc["exc"][v, :c] = rng.integers(0, 2**64, c, dtype=np.uint64).view(np.float64)  # arbitrary bit patterns incl. NaN payloads
    return enc


@pytest.mark.parametrize("placement", ["random", "front", "edges"])
@pytest.mark.parametrize("bw", [0, 6, 17, 52])
def test_exceptions_patched_after_the_stores_or_through_the_mask(ctx, oracle, bw, placement):
    """the three loops a store-decode wavefront picks from per vector (no exceptions / all values staged in LDS / values beyond the stage read from HBM:
    counts 0, 1..128, 129..1024) in every launch shape, positions on the quarter boundaries; and ALPGPU_OPT_DECODE_PATCH_AFTER under every limit — a no-op
    in the default build, the patch arms in an -DALPGPU_DECODE_PATCH_MODE=1 / 2 build selected with ALPGPU_LIB — the same bits as the oracle"""
    from alp_amd import capi
    rng = np.random.default_rng(7000 + bw)
    counts = [0, 1, 2, 3, 7, 8, 9, 20, 31, 32, 33, 63, 64, 65, 100, 128, 129, 300, 1024, 1, 64, 0, 5]
    enc = _alp_vectors_with_exception_counts(rng, counts, bw, placement)
    want = oracle.decode_column(enc)
    try:
        for limit in (64, 0, 8):
            ctx.set_option(capi.OPT_DECODE_PATCH_AFTER, limit)
            for vpw, pairing, plain in ((0, 0, 0), (1, 0, 0), (2, 0, 0), (4, 0, 0), (0, 1, 0), (0, 2, 0), (0, 3, 0), (2, 0, 1)):
                ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
                ctx.set_option(capi.OPT_DECODE_PAIRING, pairing)
                ctx.set_option(capi.OPT_DECODE_PLAIN_STORES, plain)
                got = gpu_decode(ctx, enc)
                assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), (bw, placement, limit, vpw, pairing, plain)
    finally:
        ctx.set_option(capi.OPT_DECODE_PATCH_AFTER, 0)
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)
        ctx.set_option(capi.OPT_DECODE_PAIRING, 0)
        ctx.set_option(capi.OPT_DECODE_PLAIN_STORES, 0)


def test_patched_exceptions_on_a_full_chip(ctx, oracle):
    """a column long enough to keep every CU busy for many waves of workgroups (store -> patch ordering under load): 40 000 narrow vectors with 20
    exceptions each (the bench's 2 % sweep), decoded three times with each route; the patch route against the mask route bit for bit"""
    import torch
    from alp_amd import capi
    rng = np.random.default_rng(99)
    n, c, bw = 40000, 20, 3
    enc = dict(scheme=np.full(n, 2, np.uint8), e=np.full(n, 2, np.uint8), f=np.zeros(n, np.uint8), bw=np.full(n, bw, np.uint8),
               lbw=np.zeros(n, np.uint8), base=rng.integers(-1000, 1000, n), exc_cnt=np.full(n, c, np.uint16),
               packed=np.zeros((n, 1024), np.int64), packed_left=np.zeros((n, 1024), np.uint16),
               exc=np.zeros((n, 1024), np.float64), pos=np.zeros((n, 1024), np.uint16), dict=np.zeros((n // 100, 8), np.uint16),
               dict_size=np.zeros(n // 100, np.uint8), k=np.ones(n // 100, np.uint8), combos=np.zeros((n // 100, 10), np.int32))
    enc["packed"][:, :16 * bw] = rng.integers(-2**63, 2**63 - 1, (n, 16 * bw), dtype=np.int64)
    pos = np.sort(np.argsort(rng.random((n, 1024)), axis=1)[:, :c], axis=1)
    enc["pos"][:, :c] = pos.astype(np.uint16)
    enc["exc"][:, :c] = rng.integers(0, 2**64, (n, c), dtype=np.uint64).view(np.float64)
    rg, vec, packed, exc = layout.compact(enc)
    col = capi.DeviceColumn.from_host(rg, vec, packed, exc)
    try:
        ctx.set_option(capi.OPT_DECODE_PATCH_AFTER, 0)
        ref = ctx.decode(col).clone()
        ctx.synchronize()
        # the mask route is the oracle's on a sample of the vectors
        sample = {k: (a[:300] if a.shape[0] == n else a[:3]) for k, a in enc.items()}
        assert np.array_equal(ref[:300 * 1024].cpu().numpy().view(np.uint64), oracle.decode_column(sample).view(np.uint64))
        ctx.set_option(capi.OPT_DECODE_PATCH_AFTER, 64)
        for vpw in (0, 1, 2, 0):
            ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
            out = ctx.decode(col)
            ctx.synchronize()
            assert torch.equal(out.view(torch.int64), ref.view(torch.int64)), vpw
    finally:
        ctx.set_option(capi.OPT_DECODE_PATCH_AFTER, 0)
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)


@pytest.mark.parametrize("exceptions", [0, 5, 200], ids=["clean", "staged_exceptions", "exceptions_beyond_the_stage"])
def test_shortcut_arithmetic_at_every_width(ctx, oracle, exceptions):
    """vectors that qualify for the conversion shortcut (|base + digit| < 2^51 and its product with 10^f inside int64) at EVERY bit width 0..52: widths
    <= 32 take the 32-bit unpack + {0x43380000, digit} form of the store decode (round 5), wider ones the 64-bit form, with bases on both sides of
    zero and at the shortcut's bound for the factor; random packed words; every launch shape; against the oracle's falp + patch"""
    from alp_amd import capi
    rng = np.random.default_rng(4242 + exceptions)
    bound = [2251799813685247, 2251799813685247, 2251799813685247, 2251799813685247, 922337203685477, 92233720368547, 9223372036854, 922337203685, 92233720368,
             9223372036, 922337203, 92233720, 9223372, 922337, 92233, 9223, 922, 92, 9]
    rows = []
    for bw in range(0, 53):
        for f in (0, 2, 6, 11, 14, 18):
            span = (1 << bw) - 1
            if span > bound[f]:
                continue
            for base in (0, -1, 1 - (1 << 31), (1 << 32) + 12345, -bound[f], bound[f] - span, -(bound[f] // 3)):
                if base < -bound[f] or base + span > bound[f]:
                    continue
                rows.append((bw, f, min(18, f + int(rng.integers(0, 3))), base))
    n = len(rows)
    enc = dict(scheme=np.full(n, 2, np.uint8), e=np.array([r[2] for r in rows], np.uint8), f=np.array([r[1] for r in rows], np.uint8),
               bw=np.array([r[0] for r in rows], np.uint8), lbw=np.zeros(n, np.uint8), base=np.array([r[3] for r in rows], np.int64), exc_cnt=np.zeros(n, np.uint16),
               packed=np.zeros((n, 1024), np.int64), packed_left=np.zeros((n, 1024), np.uint16), exc=np.zeros((n, 1024), np.float64), pos=np.zeros((n, 1024), np.uint16),
               dict=np.zeros(((n + 99) // 100, 8), np.uint16), dict_size=np.zeros((n + 99) // 100, np.uint8), k=np.ones((n + 99) // 100, np.uint8),
               combos=np.zeros(((n + 99) // 100, 10), np.int32))
    for v, (bw, f, e, base) in enumerate(rows):
        enc["packed"][v, :16 * bw] = rng.integers(-2**63, 2**63 - 1, 16 * bw, dtype=np.int64)
        c = exceptions if exceptions < 200 else int(rng.integers(129, 400))
        enc["exc_cnt"][v] = c
        enc["pos"][v, :c] = np.sort(rng.choice(1024, c, replace=False)).astype(np.uint16)
        enc["exc"][v, :c] = rng.integers(0, 2**64, c, dtype=np.uint64).view(np.float64)
    want = oracle.decode_column(enc)
    try:
        for vpw in (0, 1, 2, 4):
            ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
            got = gpu_decode(ctx, enc)
            bad = np.nonzero((got.view(np.uint64) != want.view(np.uint64)).reshape(n, 1024).any(axis=1))[0]
            assert bad.size == 0, (vpw, [rows[i] for i in bad[:5]])
    finally:
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)


@pytest.mark.parametrize("lead_us", [1, 40])
def test_read_ahead_does_not_change_a_byte(ctx, oracle, lead_us):
    """ALPGPU_OPT_DECODE_READ_AHEAD: the decode kernel reports its position, a second kernel on the context's second stream reads the column's streams
    ahead of it (read_ahead_kernels.hip).  A column long enough to take the option (>= 32768 vectors; widths 0..53 by rowgroup, every other rowgroup with
    exceptions), every launch shape, a lead so short that the read-ahead waits for the decode all the time and the default one; four launches in a row
    (the tag in the progress word changes per launch); against the plain decode bit for bit, and a sample of that against the oracle"""
    import torch
    from alp_amd import capi
    rng = np.random.default_rng(5)
    n = 40000
    rgs = n // 100
    bw_rg = rng.integers(0, 54, rgs)
    bw = np.repeat(bw_rg, 100).astype(np.uint8)
    c = np.repeat(np.where(np.arange(rgs) % 2 == 0, 0, 17), 100).astype(np.uint16)
    enc = dict(scheme=np.full(n, 2, np.uint8), e=np.full(n, 14, np.uint8), f=np.full(n, 12, np.uint8), bw=bw,
               lbw=np.zeros(n, np.uint8), base=rng.integers(-1000, 1000, n), exc_cnt=c,
               packed=np.zeros((n, 1024), np.int64), packed_left=np.zeros((n, 1024), np.uint16),
               exc=np.zeros((n, 1024), np.float64), pos=np.zeros((n, 1024), np.uint16), dict=np.zeros((rgs, 8), np.uint16),
               dict_size=np.zeros(rgs, np.uint8), k=np.ones(rgs, np.uint8), combos=np.zeros((rgs, 10), np.int32))
    for r in range(rgs):
        w = 16 * int(bw_rg[r])
        if w:
            enc["packed"][100 * r: 100 * r + 100, :w] = rng.integers(-2**63, 2**63 - 1, (100, w), dtype=np.int64)
    pos = np.sort(np.argsort(rng.random((n, 1024)), axis=1)[:, :17], axis=1)
    enc["pos"][:, :17] = pos.astype(np.uint16)
    enc["exc"][:, :17] = rng.integers(0, 2**64, (n, 17), dtype=np.uint64).view(np.float64)
    rg, vec, packed, exc = layout.compact(enc)
    col = capi.DeviceColumn.from_host(rg, vec, packed, exc)
    try:
        ref = ctx.decode(col).clone()
        ctx.synchronize()
        sample = {k: (a[:300] if a.shape[0] == n else a[:3]) for k, a in enc.items()}
        assert np.array_equal(ref[:300 * 1024].cpu().numpy().view(np.uint64), oracle.decode_column(sample).view(np.uint64))
        ctx.set_option(capi.OPT_DECODE_READ_AHEAD, 1)
        ctx.set_option(capi.OPT_DECODE_READ_AHEAD_US, lead_us)
        for vpw in (0, 1, 2, 0):
            ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
            out = torch.zeros_like(ref)
            ctx.decode(col, out)
            ctx.synchronize()
            assert torch.equal(out.view(torch.int64), ref.view(torch.int64)), vpw
    finally:
        ctx.set_option(capi.OPT_DECODE_READ_AHEAD, -1)
        ctx.set_option(capi.OPT_DECODE_READ_AHEAD_US, 0)
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)


def test_read_ahead_over_a_column_encoded_out_of_order(ctx):
    """records that are not in vector order (ALPGPU_OPT_ENCODE_UNORDERED): the read-ahead reads them record by record where a batch is not one span"""
    import torch
    from alp_amd import capi
    g = torch.Generator(device="cuda:0")
    g.manual_seed(11)
    n = 36000
    x = torch.round(torch.rand(n * 1024, dtype=torch.float64, device="cuda:0", generator=g) * 1e4, decimals=2)
    x[::97] = torch.rand(x[::97].shape, dtype=torch.float64, device="cuda:0", generator=g)  # exceptions
    x[n // 2 * 1024:] = torch.rand(n * 1024 - n // 2 * 1024, dtype=torch.float64, device="cuda:0", generator=g)  # the second half: full-precision values -> ALP_RD rowgroups (right + left words, 2-byte exception values)
    try:
        ctx.set_option(capi.OPT_ENCODE_UNORDERED, 1)
        col = ctx.encode(x)
        ctx.column_totals(col)
        ctx.set_option(capi.OPT_DECODE_READ_AHEAD, 1)
        out = ctx.decode(col)
        ctx.synchronize()
        assert torch.equal(out.view(torch.int64), x.view(torch.int64))
        ctx.set_option(capi.OPT_ENCODE_UNORDERED, 0)  # ... and the same column in vector order (ALP and ALP_RD records in one span per batch)
        col = ctx.encode(x)
        ctx.column_totals(col)
        assert ctx.decode_reads_ahead(col)
        out = ctx.decode(col)
        ctx.synchronize()
        assert torch.equal(out.view(torch.int64), x.view(torch.int64))
    finally:
        ctx.set_option(capi.OPT_ENCODE_UNORDERED, 0)
        ctx.set_option(capi.OPT_DECODE_READ_AHEAD, -1)


@pytest.mark.parametrize("exc", [0, 20])
def test_long_columns_of_narrow_vectors_take_the_read_ahead_by_themselves(ctx, oracle, exc):
    """the default (ALPGPU_OPT_DECODE_READ_AHEAD = -1): a column of >= 262144 vectors of <= 7 bits runs with the read-ahead and the launch shape that goes
    with it (one vector per workgroup without exceptions, two with); a wider or shorter one does not.  Same bytes as with the option off, and a sample of
    them against the oracle's falp + patch"""
    import torch
    import bench
    from alp_amd import capi
    n = 270000
    col, vec, _ = bench.build_decode_column(n, 0, seed=3, bw_of_rowgroup=3, exc_per_vec=exc)
    wide, _, _ = bench.build_decode_column(n, 0, seed=3, bw_of_rowgroup=20, exc_per_vec=exc)
    short, _, _ = bench.build_decode_column(140000, 0, seed=3, bw_of_rowgroup=3, exc_per_vec=exc)
    try:
        assert ctx.decode_vectors_per_wg(col) == (2 if exc else 1)
        assert ctx.decode_vectors_per_wg(short) == 2 and ctx.decode_vectors_per_wg(wide) == (2 if exc else 1)  # (20 bits: one per workgroup from 17.5 bits on without exceptions)
        got = ctx.decode(col).clone()
        again = ctx.decode(col).clone()  # (the progress word's tag changes per launch)
        ctx.set_option(capi.OPT_DECODE_READ_AHEAD, 0)
        assert ctx.decode_vectors_per_wg(col) == 2
        ref = ctx.decode(col)
        ctx.synchronize()
        assert torch.equal(got.view(torch.int64), ref.view(torch.int64)) and torch.equal(again.view(torch.int64), ref.view(torch.int64))
        # the oracle on the first and the last 200 vectors
        for lo in (0, n - 200):
            idx = np.arange(lo, lo + 200)
            sub = {k: vec[k][idx].copy() for k in ("bw", "e", "f", "base", "exc_cnt", "lbw")}
            sub["scheme"] = vec["scheme"][idx].astype(np.uint8)
            packed = np.zeros((200, 1024), np.uint64)
            p8 = packed.view(np.uint8).reshape(200, 8192)
            rec = (10 * exc + 7) // 8 * 8
            sub["exc"] = np.zeros((200, 1024), np.float64)
            sub["pos"] = np.zeros((200, 1024), np.uint16)
            for i, v in enumerate(idx):
                o = int(vec["packed_off"][v])
                p8[i, : 128 * 3] = col.packed[o:o + 128 * 3].cpu().numpy()
                if exc:
                    r = col.exc[int(vec["exc_off"][v]): int(vec["exc_off"][v]) + rec].cpu().numpy()
                    sub["exc"][i, :exc] = r[: 8 * exc].view(np.float64)
                    sub["pos"][i, :exc] = r[8 * exc: 10 * exc].view(np.uint16)
            sub["packed"] = packed
            sub["packed_left"] = np.zeros((200, 1024), np.uint16)
            sub["dict"] = np.zeros((3, 8), np.uint16)
            sub["dict_size"] = np.zeros(3, np.uint8)
            want = oracle.decode_column(sub)
            assert np.array_equal(ref.view(-1, 1024)[lo:lo + 200].cpu().numpy().view(np.uint64).reshape(-1), want.view(np.uint64).reshape(-1))
    finally:
        ctx.set_option(capi.OPT_DECODE_READ_AHEAD, -1)


def test_a_column_whose_regions_differ_is_decoded_region_by_region(ctx, oracle):
    """ALPGPU_OPT_DECODE_SEGMENTS: after alpgpu_column_totals the context knows the column's segments; a column of three kinds of regions (3-bit vectors with 20
    exceptions, 44-bit vectors, 12-bit vectors; the middle one short) is decoded in three launches with three shapes, a uniform one in one; every launch plan
    writes the same bytes (the option off, shapes forced, the blob route through alpgpu_column_from_blob on a second context), and a re-encode forgets the plan"""
    import torch
    import bench
    from alp_amd import capi
    n = 300000
    idx = np.arange(n)
    a, b = 140000, 200000  # (not multiples of the plan's segments — 32 800 vectors here: the segments decide where runs begin)
    bw = np.where(idx < a, 3, np.where(idx < b, 44, 12))
    exc = np.where(idx < a, 20, 0)
    col, vec, _ = bench.build_decode_column(n, 0, seed=5, bw_of_rowgroup=bw, exc_per_vec=exc)
    uni, _, _ = bench.build_decode_column(n, 0, seed=5, bw_of_rowgroup=12, exc_per_vec=0)
    try:
        assert ctx.decode_runs(col) == 1  # nobody has looked at the column yet
        ref = ctx.decode(col).clone()
        ctx.synchronize()
        ctx.column_totals(col)
        ctx.column_totals(uni)
        assert ctx.decode_runs(uni) == 1
        runs = ctx.decode_runs(col)
        assert 3 <= runs <= 5, runs  # three kinds; the segments that straddle a border may form runs of their own
        for _ in range(2):
            out = torch.zeros_like(ref)
            ctx.decode(col, out)
            ctx.synchronize()
            assert torch.equal(out.view(torch.int64), ref.view(torch.int64))
        ctx.set_option(capi.OPT_DECODE_SEGMENTS, 0)
        assert ctx.decode_runs(col) == 1
        ctx.set_option(capi.OPT_DECODE_SEGMENTS, 1)
        # the plan belongs to THIS column: the next column a caching allocator puts into the same buffers has other stream sizes
        kept = int(col.c.packed_bytes_hint)
        col.c.packed_bytes_hint = kept + 128
        assert ctx.decode_runs(col) == 1
        col.c.packed_bytes_hint = kept
        assert ctx.decode_runs(col) == runs
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 2)  # a forced shape is a forced shape
        assert ctx.decode_runs(col) == 1
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)
        # the oracle on vectors around both borders
        for lo in (a - 100, b - 100):
            sel = np.arange(lo, lo + 200)
            sub = {k: vec[k][sel].copy() for k in ("bw", "e", "f", "base", "exc_cnt", "lbw")}
            sub["scheme"] = vec["scheme"][sel].astype(np.uint8)
            packed = np.zeros((200, 1024), np.uint64)
            p8 = packed.view(np.uint8).reshape(200, 8192)
            sub["exc"] = np.zeros((200, 1024), np.float64)
            sub["pos"] = np.zeros((200, 1024), np.uint16)
            rec = (10 * 20 + 7) // 8 * 8
            for i, v in enumerate(sel):
                o, w = int(vec["packed_off"][v]), int(vec["bw"][v])
                p8[i, : 128 * w] = col.packed[o:o + 128 * w].cpu().numpy()
                if vec["exc_cnt"][v]:
                    r = col.exc[int(vec["exc_off"][v]): int(vec["exc_off"][v]) + rec].cpu().numpy()
                    sub["exc"][i, :20] = r[:160].view(np.float64)
                    sub["pos"][i, :20] = r[160:200].view(np.uint16)
            sub["packed"] = packed
            sub["packed_left"] = np.zeros((200, 1024), np.uint16)
            sub["dict"] = np.zeros((3, 8), np.uint16)
            sub["dict_size"] = np.zeros(3, np.uint8)
            want = oracle.decode_column(sub)
            assert np.array_equal(ref.view(-1, 1024)[lo:lo + 200].cpu().numpy().view(np.uint64).reshape(-1), want.view(np.uint64).reshape(-1))
        # the blob route: a second context learns the segments from the blob's descriptors
        blob = ctx.to_blob(col, n * 1024)
        other = capi.Context(0)
        col2, n_values = other.from_blob(blob)
        assert n_values == n * 1024 and other.decode_runs(col2) == runs
        out2 = other.decode(col2)
        other.synchronize()
        assert torch.equal(out2.view(torch.int64), ref.view(torch.int64))
    finally:
        ctx.set_option(capi.OPT_DECODE_SEGMENTS, 1)
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)
