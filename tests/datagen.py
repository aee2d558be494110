"""Seeded synthetic double columns shared by tests and bench.py.  numpy only."""
import numpy as np

VEC = 1024


def decimal_column(n_vectors, decimals=2, lo=-1e5, hi=1e5, seed=42):
    rng = np.random.default_rng(seed)
    x = rng.uniform(lo, hi, n_vectors * VEC)
    return np.round(x, decimals)


def mixed_column(n_vectors, seed=42, exc_rate=0.01, special_rate=0.001, decimals_per_rowgroup=(1, 2, 4)):
    """round(x, d) with d cycling per rowgroup, a fraction of full-precision values (exceptions) and specials."""
    rng = np.random.default_rng(seed)
    n = n_vectors * VEC
    x = rng.uniform(-1e5, 1e5, n)
    out = np.empty(n, np.float64)
    for r in range((n_vectors + 99) // 100):
        s = slice(r * 100 * VEC, min(n, (r + 1) * 100 * VEC))
        out[s] = np.round(x[s], decimals_per_rowgroup[r % len(decimals_per_rowgroup)])
    m = rng.random(n) < exc_rate
    out[m] = x[m] * np.pi
    sp = rng.random(n) < special_rate
    specials = np.array([np.nan, np.inf, -np.inf, -0.0])
    out[sp] = specials[rng.integers(0, 4, int(sp.sum()))]
    return out


def rd_column(n_vectors, seed=42, kind="unit"):
    rng = np.random.default_rng(seed)
    n = n_vectors * VEC
    if kind == "unit":
        return rng.random(n)
    return rng.uniform(-90.0, 90.0, n)  # lat/lon-like, full precision


def drifting_column(n_vectors, seed=7):
    """precision changes inside a rowgroup -> several (e,f) candidates (k > 1): exercises second-level sampling"""
    rng = np.random.default_rng(seed)
    n = n_vectors * VEC
    x = rng.uniform(0, 1000, n)
    out = np.empty(n, np.float64)
    for v in range(n_vectors):
        d = (1, 3, 5, 2)[(v // 7) % 4]
        out[v * VEC:(v + 1) * VEC] = np.round(x[v * VEC:(v + 1) * VEC], d)
    return out


def adversarial_vectors():
    """hand-made single vectors for the corner cases of encode (all exceptions, exceptions 0..1022, ...)"""
    base = np.round(np.random.default_rng(3).uniform(0, 100, VEC), 2)
    cases = {}
    cases["plain"] = base.copy()
    a = base.copy(); a[:] = np.pi * (np.arange(VEC) + 1); cases["all_exceptions"] = a
    a = base.copy(); a[:1023] = np.e * (np.arange(1023) + 1); cases["exceptions_0_to_1022"] = a
    a = base.copy(); a[:5] = np.nan; cases["prefix_nan"] = a
    a = base.copy(); a[1023] = np.inf; a[0] = -np.inf; cases["inf_ends"] = a
    a = base.copy(); a[::2] = -0.0; cases["half_negzero"] = a
    a = np.zeros(VEC); cases["all_zero"] = a
    a = np.full(VEC, 10.23); cases["constant"] = a
    a = base.copy(); a[7] = 2.0**63; a[8] = -2.0**63; a[9] = 1e300; a[10] = 5e-324; cases["huge_tiny"] = a
    a = base.copy(); a[100:200] = np.random.default_rng(4).random(100); cases["exception_block"] = a
    return cases


def search_boundary_values():
    """doubles on the decision boundaries of the (e,f) search's arithmetic: v * 10^e around 2^63 and 2^64 (the int64 product of
    the decoder wraps there, the encoder switches to its sentinel), the values the sentinel 2^63 - 1024 decodes to, encoded
    integers r with r * 10^f around 2^63, signed zeros and the smallest magnitudes.  Used one value per (constant) vector, so the
    rowgroup state shows the first (e,f) candidate that round-trips it (or ALP_RD if none does)."""
    import math
    vals = [0.0, -0.0, 5e-324, -5e-324, 2.2250738585072014e-308, math.inf, -math.inf, math.nan, 1.7976931348623157e308]
    sentinel = 9223372036854774784.0
    for e in range(19):
        fe = float(f"1e-{e}")
        for big in (sentinel, 2.0**63, 2.0**63 + 2048.0, 2.0**64, 2.0**64 - 2048.0, 2.0**62, 2.0**53, 2.0**51):
            for base in (big * fe, big / 10.0**e):
                x = base
                vals += [x, -x]
                up = dn = x
                for _ in range(2):
                    up, dn = math.nextafter(up, math.inf), math.nextafter(dn, -math.inf)
                    vals += [up, dn, -up]
        for f in range(e + 1):
            for d in (-1, 0, 1):
                r = (1 << 63) // 10**f + d
                vals += [(r * 10**f) / 10**e, -(r * 10**f) / 10**e]
        # half-integers between 2^51 and 2^52: negative ones survive the magic-number rounding with their half (t + M < 2^52)
        for k2 in (2**52 + 1, 2**52 + 3, 2**53 - 1, 2**53 - 3, 3 * 2**51 + 1, 2**52 - 1, 2**53 + 2):  # k2 / 2 = the value of t aimed at
            vals += [k2 / (2 * 10**e), -k2 / (2 * 10**e), (k2 / 2.0) * fe, -(k2 / 2.0) * fe]
    out = np.array(vals, np.float64)
    return out


def search_boundary_mixtures(seed, n_vectors=4):
    """vectors whose 32 sampled positions (every 32nd value) hold boundary values mixed with two-decimal values, so that the
    candidates' exception counts and ranges — not just the first that fits — decide the rowgroup state"""
    rng = np.random.default_rng(seed)
    pool = search_boundary_values()
    pool = pool[np.isfinite(pool)]
    x = np.round(rng.uniform(-1000, 1000, n_vectors * VEC), 2)
    for v in range(n_vectors):
        k = int(rng.integers(1, 12))
        pos = rng.choice(32, k, replace=False) * 32 + v * VEC
        x[pos] = rng.choice(pool, k)
    return x


# ---- single precision --------------------------------------------------------------------
def decimal_column_f32(n_vectors, decimals=2, lo=0.0, hi=1000.0, seed=42):
    rng = np.random.default_rng(seed)
    return np.round(rng.uniform(lo, hi, n_vectors * VEC), decimals).astype(np.float32)


def mixed_column_f32(n_vectors, seed=42, exc_rate=0.01, special_rate=0.001, decimals_per_rowgroup=(1, 2, 3)):
    rng = np.random.default_rng(seed)
    n = n_vectors * VEC
    x = rng.uniform(-1000.0, 1000.0, n)
    out = np.empty(n, np.float32)
    for r in range((n_vectors + 99) // 100):
        s = slice(r * 100 * VEC, min(n, (r + 1) * 100 * VEC))
        out[s] = np.round(x[s], decimals_per_rowgroup[r % len(decimals_per_rowgroup)]).astype(np.float32)
    m = rng.random(n) < exc_rate
    out[m] = (x[m] * np.pi).astype(np.float32)
    sp = rng.random(n) < special_rate
    specials = np.array([np.nan, np.inf, -np.inf, -0.0], np.float32)
    out[sp] = specials[rng.integers(0, 4, int(sp.sum()))]
    return out


def rd_column_f32(n_vectors, seed=42, kind="unit"):
    rng = np.random.default_rng(seed)
    n = n_vectors * VEC
    if kind == "unit":
        return rng.random(n, dtype=np.float32)
    return rng.uniform(-90.0, 90.0, n).astype(np.float32)


def drifting_column_f32(n_vectors, seed=7):
    rng = np.random.default_rng(seed)
    n = n_vectors * VEC
    x = rng.uniform(0, 100, n)
    out = np.empty(n, np.float32)
    for v in range(n_vectors):
        d = (1, 3, 4, 2)[(v // 7) % 4]
        out[v * VEC:(v + 1) * VEC] = np.round(x[v * VEC:(v + 1) * VEC], d).astype(np.float32)
    return out


def adversarial_vectors_f32():
    base = np.round(np.random.default_rng(3).uniform(0, 100, VEC), 2).astype(np.float32)
    cases = {}
    cases["plain"] = base.copy()
    a = base.copy(); a[:] = (np.pi * (np.arange(VEC) + 1)).astype(np.float32); cases["all_exceptions"] = a
    a = base.copy(); a[:1023] = (np.e * (np.arange(1023) + 1)).astype(np.float32); cases["exceptions_0_to_1022"] = a
    a = base.copy(); a[:5] = np.nan; cases["prefix_nan"] = a
    a = base.copy(); a[1023] = np.inf; a[0] = -np.inf; cases["inf_ends"] = a
    a = base.copy(); a[::2] = -0.0; cases["half_negzero"] = a
    cases["all_zero"] = np.zeros(VEC, np.float32)
    cases["constant"] = np.full(VEC, 10.23, np.float32)
    a = base.copy(); a[7] = 2.0**31; a[8] = -2.0**31; a[9] = 1e30; a[10] = 1e-45; a[11] = 2.0**63; a[12] = -2.0**63
    a[13] = 2147483520.0; a[14] = -2147483904.0; a[15] = 1.1754942e-38; a[16] = 16777216.0; a[17] = 8388609.0; a[18] = 4194303.5
    cases["huge_tiny"] = a
    a = base.copy(); a[100:200] = np.random.default_rng(4).random(100, dtype=np.float32); cases["exception_block"] = a
    a = np.random.default_rng(5).integers(-2**31, 2**31, VEC).astype(np.float32); cases["big_integers"] = a
    a = (np.random.default_rng(6).integers(-2**24, 2**24, VEC)).astype(np.float32); cases["exact_integers"] = a
    a = np.random.default_rng(7).integers(0, 1 << 23, VEC).astype(np.uint32).view(np.float32).copy(); cases["denormals"] = a
    return cases


def every_bit_width_column(n_vectors=208, seed=77, exceptions=True):
    """ALP vectors of every packed width 0..64 inside ALP rowgroups.  The sampled vectors (index % 12 == 0 inside a rowgroup) hold
    2^62 + 1024 m, m < 2^30: only (e,f) = (0,0) encodes them (times ten they leave int64) and their range needs 40 bits, so the rowgroup is
    ALP with k = 1 and (0,0).  Under (0,0) every integer-valued double below 2^63 encodes exactly, so the other vectors choose their width:
    vector j gets width (j mod 65) from integers B + m (m < 2^w, w <= 52) or multiples of 2^(w-52) spanning 2^w (w >= 53), extremes present;
    with `exceptions`, every third vector also carries NaN / fraction / -0.0 values (exceptions, and the filler at their slots)."""
    rng = np.random.default_rng(seed)
    col = np.empty(n_vectors * 1024, np.float64)
    for v in range(n_vectors):
        o = v * 1024
        if (v % 100) % 12 == 0:
            col[o:o + 1024] = 2.0**62 + 1024.0 * rng.integers(0, 2**30, 1024).astype(np.float64)
            continue
        w = v % 65
        if w == 0:
            vals = np.full(1024, float(rng.integers(-2**40, 2**40)))
        elif w <= 52:
            base = int(rng.integers(-2**52, 2**52 - 2**w)) if w < 52 else -2**51
            m = rng.integers(0, 2**w, 1024, dtype=np.uint64).astype(np.int64) if w < 63 else None
            m[0], m[1] = 0, 2**w - 1
            vals = (base + m).astype(np.float64)
        else:
            step = 2 ** (w - 52)
            m = rng.integers(-2**51 + 1, 2**51, 1024)
            m[0], m[1] = -2**51 + 1, 2**51 - 1  # range 2^52 - 2 steps: needs w bits (2^(w-1) < (2^52 - 2) * step < 2^w)
            vals = m.astype(np.float64) * float(step)
        rng.shuffle(vals)
        if exceptions and v % 3 == 1:
            idx = rng.choice(1024, int(rng.integers(1, 40)), replace=False)
            vals[idx] = rng.choice([np.nan, 0.5, -0.0, 1e300, -2.5e-7], idx.size)
        col[o:o + 1024] = vals
    return col
