"""GPU parity of the batch vector primitives (the reference's per-vector API, n vectors per call) against the
oracle: ffor/unffor for EVERY bit width (u64 0..64, u16 0..16), falp, decode, patch, encode_simdized, encode with
second-level sampling, analyze_ffor, rd encode/decode."""
import numpy as np
import pytest
import torch

import datagen

pytestmark = pytest.mark.gpu


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_ffor_unffor_falp_u64_every_bit_width(ctx, oracle):
    rng = np.random.default_rng(5)
    bws = np.arange(0, 65, dtype=np.uint8)
    n = bws.size
    base = rng.integers(-2**62, 2**62, n)
    vals = np.zeros((n, 1024), np.uint64)
    for i, bw in enumerate(bws):
        span = (1 << int(bw)) - 1 if bw < 64 else (1 << 64) - 1
        r = (rng.integers(0, 2**63, 1024, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, 1024, dtype=np.uint64)) & np.uint64(span)
        vals[i] = r + np.uint64(int(base[i]) & (2**64 - 1))
    want_packed = np.stack([oracle.ffor_u64(vals[i], int(bws[i]), int(base[i])) for i in range(n)])
    d_packed = torch.zeros((n, 1024), dtype=torch.int64, device="cuda")
    ctx.ffor_i64(cu(vals.view(np.int64)), d_packed, cu(bws), cu(base))
    ctx.synchronize()
    got = d_packed.cpu().numpy().view(np.uint64)
    for i, bw in enumerate(bws):
        assert np.array_equal(got[i, :16 * int(bw)], want_packed[i, :16 * int(bw)]), f"ffor bw={bw}"
        assert not got[i, 16 * int(bw):].any(), f"ffor bw={bw} wrote past 16*bw words"
    d_out = torch.zeros((n, 1024), dtype=torch.int64, device="cuda")
    ctx.unffor_i64(cu(want_packed.view(np.int64)), d_out, cu(bws), cu(base))
    ctx.synchronize()
    assert np.array_equal(d_out.cpu().numpy().view(np.uint64), vals), "unffor"
    fac = rng.integers(0, 19, n).astype(np.uint8)
    exp = np.maximum(fac, rng.integers(0, 19, n)).astype(np.uint8)
    d_dbl = torch.zeros((n, 1024), dtype=torch.float64, device="cuda")
    ctx.falp(cu(want_packed.view(np.int64)), d_dbl, cu(bws), cu(base), cu(fac), cu(exp))
    ctx.synchronize()
    want = np.stack([oracle.falp(want_packed[i], int(bws[i]), int(base[i]), int(fac[i]), int(exp[i])) for i in range(n)])
    assert np.array_equal(d_dbl.cpu().numpy().view(np.uint64), want.view(np.uint64)), "falp"
    # decoder::decode on the unpacked integers gives the same doubles (falp == unffor + decode)
    d_dbl2 = torch.zeros((n, 1024), dtype=torch.float64, device="cuda")
    ctx.decode_values(cu(vals.view(np.int64)), d_dbl2, cu(fac), cu(exp))
    ctx.synchronize()
    assert np.array_equal(d_dbl2.cpu().numpy().view(np.uint64), want.view(np.uint64)), "decode"


def test_ffor_unffor_u16_every_bit_width(ctx, oracle):
    rng = np.random.default_rng(6)
    bws = np.arange(0, 17, dtype=np.uint8)
    n = bws.size
    vals = np.stack([(rng.integers(0, 1 << 16, 1024) & ((1 << int(b)) - 1)).astype(np.uint16) for b in bws])
    want = np.stack([oracle.ffor_u16(vals[i], int(bws[i])) for i in range(n)])
    d_packed = torch.zeros((n, 1024), dtype=torch.int16, device="cuda")
    ctx.ffor_u16(cu(vals.view(np.int16)), d_packed, cu(bws))
    ctx.synchronize()
    got = d_packed.cpu().numpy().view(np.uint16)
    for i, bw in enumerate(bws):
        assert np.array_equal(got[i, :64 * int(bw)], want[i, :64 * int(bw)]), f"ffor u16 bw={bw}"
    d_out = torch.zeros((n, 1024), dtype=torch.int16, device="cuda")
    ctx.unffor_u16(cu(want.view(np.int16)), d_out, cu(bws))
    ctx.synchronize()
    assert np.array_equal(d_out.cpu().numpy().view(np.uint16), vals)


def test_ffor_unffor_u8_every_bit_width(ctx, oracle):
    rng = np.random.default_rng(7)
    bws = np.arange(0, 9, dtype=np.uint8)
    n = bws.size
    base = rng.integers(0, 256, n).astype(np.uint8)
    vals = np.stack([(((rng.integers(0, 256, 1024) & ((1 << int(b)) - 1)) + int(base[i])) & 0xFF).astype(np.uint8) for i, b in enumerate(bws)])
    want = np.stack([oracle.ffor_u8(vals[i], int(bws[i]), int(base[i])) for i in range(n)])
    d_packed = torch.zeros((n, 1024), dtype=torch.uint8, device="cuda")
    ctx.ffor_u8(cu(vals), d_packed, cu(bws), cu(base))
    ctx.synchronize()
    got = d_packed.cpu().numpy()
    for i, bw in enumerate(bws):
        assert np.array_equal(got[i, :128 * int(bw)], want[i, :128 * int(bw)]), f"ffor u8 bw={bw}"
        assert not got[i, 128 * int(bw):].any()
    d_out = torch.zeros((n, 1024), dtype=torch.uint8, device="cuda")
    ctx.unffor_u8(cu(want), d_out, cu(bws), cu(base))
    ctx.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), vals)


def test_encode_simdized_analyze_patch(ctx, oracle):
    cases = datagen.adversarial_vectors()
    efs = [(14, 12), (18, 18), (0, 0), (5, 2), (16, 0)]
    vecs, fac, exp = [], [], []
    for name, v in cases.items():
        for e, f in efs:
            vecs.append(v), fac.append(f), exp.append(e)
    x = np.stack(vecs)
    n = x.shape[0]
    d_exc = torch.zeros((n, 1024), dtype=torch.float64, device="cuda")
    d_pos = torch.zeros((n, 1024), dtype=torch.int16, device="cuda")
    d_cnt = torch.zeros(n, dtype=torch.int16, device="cuda")
    d_enc = torch.zeros((n, 1024), dtype=torch.int64, device="cuda")
    d_fac, d_exp = cu(np.array(fac, np.uint8)), cu(np.array(exp, np.uint8))
    ctx.encode_simdized(cu(x), d_exc, d_pos, d_cnt, d_enc, d_fac, d_exp)
    d_bw = torch.zeros(n, dtype=torch.uint8, device="cuda")
    d_base = torch.zeros(n, dtype=torch.int64, device="cuda")
    ctx.analyze_ffor(d_enc, d_bw, d_base)
    d_dec = torch.zeros((n, 1024), dtype=torch.float64, device="cuda")
    ctx.decode_values(d_enc, d_dec, d_fac, d_exp)
    ctx.patch(d_dec, d_exc, d_pos, d_cnt)
    ctx.synchronize()
    enc, exc, pos = d_enc.cpu().numpy(), d_exc.cpu().numpy(), d_pos.cpu().numpy().view(np.uint16)
    cnt = d_cnt.cpu().numpy().view(np.uint16)
    for i in range(n):
        we, wx, wp, wc = oracle.encode_simdized(x[i], fac[i], exp[i])
        assert cnt[i] == wc and np.array_equal(enc[i], we), i
        assert np.array_equal(pos[i, :wc], wp[:wc]) and np.array_equal(exc[i, :wc].view(np.uint64), wx[:wc].view(np.uint64)), i
        assert (int(d_bw[i]), int(d_base[i])) == oracle.analyze_ffor(we), i
    assert np.array_equal(d_dec.cpu().numpy().view(np.uint64), x.view(np.uint64)), "decode+patch must reproduce the input bits"


def test_encode_values_second_level_sampling(ctx, oracle):
    """encoder::encode with k > 1: state from the oracle, one state per vector via state_idx"""
    import layout
    col = datagen.drifting_column(200, seed=7)
    want = oracle.encode_column(col)
    states, _, _, _ = layout.compact(want)
    assert (states["k"] > 1).any()
    n = 200
    idx = (np.arange(n) // 100).astype(np.uint32)
    d_exc = torch.zeros((n, 1024), dtype=torch.float64, device="cuda")
    d_pos = torch.zeros((n, 1024), dtype=torch.int16, device="cuda")
    d_cnt = torch.zeros(n, dtype=torch.int16, device="cuda")
    d_enc = torch.zeros((n, 1024), dtype=torch.int64, device="cuda")
    d_fac = torch.zeros(n, dtype=torch.uint8, device="cuda")
    d_exp = torch.zeros(n, dtype=torch.uint8, device="cuda")
    ctx.encode_values(cu(col.reshape(n, 1024)), cu(states.view(np.uint8)), cu(idx.view(np.int32)), d_exc, d_pos, d_cnt, d_enc, d_fac, d_exp)
    ctx.synchronize()
    assert np.array_equal(d_fac.cpu().numpy(), want["f"]) and np.array_equal(d_exp.cpu().numpy(), want["e"])
    assert np.array_equal(d_cnt.cpu().numpy().view(np.uint16), want["exc_cnt"])


def test_rd_encode_decode_vectors(ctx, oracle):
    import layout
    col = datagen.rd_column(120, seed=5)
    want = oracle.encode_column(col)
    states, _, _, _ = layout.compact(want)
    n = 120
    d_exc = torch.zeros((n, 1024), dtype=torch.int16, device="cuda")
    d_pos = torch.zeros((n, 1024), dtype=torch.int16, device="cuda")
    d_cnt = torch.zeros(n, dtype=torch.int16, device="cuda")
    d_right = torch.zeros((n, 1024), dtype=torch.int64, device="cuda")
    d_left = torch.zeros((n, 1024), dtype=torch.int16, device="cuda")
    x = cu(col.reshape(n, 1024))
    st = cu(states.view(np.uint8))
    ctx.rd_encode_vectors(x, st, None, d_exc, d_pos, d_cnt, d_right, d_left)
    d_out = torch.zeros((n, 1024), dtype=torch.float64, device="cuda")
    ctx.rd_decode_vectors(d_out, d_right, d_left, st, None, d_exc, d_pos, d_cnt)
    ctx.synchronize()
    assert np.array_equal(d_cnt.cpu().numpy().view(np.uint16), want["exc_cnt"])
    right = d_right.cpu().numpy().view(np.uint64)
    for v in range(n):
        rbw = int(want["bw"][v])
        assert np.array_equal(right[v], col.view(np.uint64)[v * 1024:(v + 1) * 1024] & np.uint64((1 << rbw) - 1))
        c = int(want["exc_cnt"][v])
        assert np.array_equal(d_pos[v, :c].cpu().numpy().view(np.uint16), want["pos"][v, :c])
        assert np.array_equal(d_exc[v, :c].cpu().numpy().view(np.uint16), want["exc"][v].view(np.uint16)[:c])
    assert torch.equal(d_out.view(torch.int64), x.view(torch.int64))
