#!/usr/bin/env python3
"""fuzz_options.py [rounds] [seed]: random COMBINATIONS of the context's options over random columns (tests/test_fuzz_gpu.py's generator; double and float).
Every option has a parity test of its own; this looks for combinations that interact.  Per round: the column is encoded by the default route (the bytes of
record), then under a random set of encode options — ordered routes must give the same streams byte for byte, the unordered route the same records
(descriptors but for their offsets) — and each column is decoded / summed / counted under a random set of decode options, with and without size hints:
decoded bytes = the input, sums and counts = the default route's.  Prints one line per failure and a summary; exit status 1 on any failure."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from alp_amd import capi  # noqa: E402
from test_fuzz_gpu import fuzz_column  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
budget_s = float(os.environ.get("FUZZ_SECONDS", "1e9"))
ctx = capi.Context(0)

ENC_OPTS = {  # option -> values a round may pick (first = default)
    capi.OPT_ENCODE_TWO_PASS: (0, 1),
    capi.OPT_ENCODE_KERNEL: (0, 1),
    capi.OPT_ENCODE_ASYNC_INIT: (1, 0, 2),
    capi.OPT_ENCODE_UNORDERED: (0, 1),
    capi.OPT_DEBUG_FORCE_STALL: (0, 0, 0, 1),
}
DEC_OPTS = {
    capi.OPT_DECODE_VECTORS_PER_WG: (0, 1, 2, 4),
    capi.OPT_DECODE_PLAIN_STORES: (0, 1),
    capi.OPT_DECODE_PAIRING: (0, 1, 2, 3),
    capi.OPT_DECODE_RESIDENCY_PAD: (-1, 0, 6, 14, 60),
    capi.OPT_DECODE_READ_AHEAD: (-1, 0, 1),
    capi.OPT_DECODE_READ_AHEAD_US: (0, 1, 25, 400),
    capi.OPT_DECODE_SEGMENTS: (1, 0),
    capi.OPT_DECODE_UNHINTED: (1, 0, 2),
    capi.OPT_CONSUMER_PIPELINED: (0, 1, 2, 3),
}
F32_VPW = (0, 1, 2, 4, 8, 16, 20, 24, 27, 30)


def set_all(opts, values):
    for o, v in zip(opts, values):
        ctx.set_option(o, v)


def defaults(opts):
    set_all(opts, [opts[o][0] for o in opts])


def records(col):
    """the column's content independent of WHERE its records lie: per vector (descriptor fields, packed bytes, exception record)"""
    rg, vec, packed, exc = col.to_host()
    out = []
    for d in vec:
        is_alp = int(d["scheme"]) == capi.SCHEME_ALP
        pb = 128 * (int(d["bw"]) + (0 if is_alp else int(d["lbw"])))
        vb = (8 if col.dtype == "f64" else 4) if is_alp else 2
        eb = ((vb + 2) * int(d["exc_cnt"]) + 7) // 8 * 8
        out.append((int(d["bw"]), int(d["e"]), int(d["f"]), int(d["lbw"]), int(d["base"]), int(d["exc_cnt"]), int(d["scheme"]),
                    packed[int(d["packed_off"]): int(d["packed_off"]) + pb].tobytes(), exc[int(d["exc_off"]): int(d["exc_off"]) + eb].tobytes()))
    return rg.tobytes(), out


def same_or_nan(a, b):
    return bool(((a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))).all())


fails = 0
t0 = time.time()
done = 0
for r in range(rounds):
    if time.time() - t0 > budget_s:
        break
    rng = np.random.default_rng(31000 + seed0 * 100003 + r)
    f64 = bool(rng.integers(0, 2))
    col_np = fuzz_column(rng, np.float64 if f64 else np.float32)
    if rng.integers(0, 6) == 0:  # sometimes long enough for the unhinted route, the read-ahead and the regions (tiled: the rowgroup decisions repeat)
        unit = col_np[: (col_np.size // 102400) * 102400] if col_np.size >= 102400 else np.resize(col_np, 102400)
        col_np = np.tile(unit, max(1, int(rng.integers(70000, 300000)) // (unit.size // 1024)))
    x = torch.from_numpy(col_np).cuda()
    it = torch.int64 if f64 else torch.int32
    n = col_np.size // 1024
    dtype = "f64" if f64 else "f32"
    defaults(ENC_OPTS), defaults(DEC_OPTS)
    ref = capi.DeviceColumn(n, dtype=dtype)
    ctx.encode(x, ref)
    ctx.synchronize()
    assert ctx.column_totals(ref)[2] == 0
    ref_streams = [a.view(np.uint8).copy() for a in ref.to_host()]
    ref_sums = ctx.decode_sum(ref).cpu().numpy()
    ref_cnt = ctx.decode_count_range(ref, -1.0e3, 1.0e3).cpu().numpy()
    # ---- a random encode route ----
    ev = [int(rng.choice(ENC_OPTS[o])) for o in ENC_OPTS]
    set_all(ENC_OPTS, ev)
    col = capi.DeviceColumn(n, dtype=dtype)
    what = f"round {r} seed {seed0} {dtype} n={n} enc={ev}"
    try:
        ctx.encode(x, col)
        ctx.synchronize()
        hinted = bool(rng.integers(0, 2))
        if hinted:
            assert ctx.column_totals(col)[2] == 0
        unordered = ev[list(ENC_OPTS).index(capi.OPT_ENCODE_UNORDERED)] == 1
        got = [a.view(np.uint8) for a in col.to_host()]
        if not unordered:
            for a, b, name in zip(got, ref_streams, ("rowgroup states", "descriptors", "packed stream", "exception stream")):
                if not np.array_equal(a, b):
                    fails += 1
                    print(f"FAIL {what}: {name} differ from the default route's")
        else:
            if n <= 20000 and records(col) != records(ref):  # (a Python loop per vector: short columns only; long ones are covered by the round trip)
                fails += 1
                print(f"FAIL {what}: the unordered column's records differ")
        # ---- random decode options over it ----
        for k in range(2):
            dv = [int(rng.choice(DEC_OPTS[o])) for o in DEC_OPTS]
            if not f64:
                dv[0] = int(rng.choice(F32_VPW))
            set_all(DEC_OPTS, dv)
            tag = f"{what} hinted={hinted} dec={dv}"
            out = ctx.decode(col)
            sums = ctx.decode_sum(col).cpu().numpy()
            cnt = ctx.decode_count_range(col, -1.0e3, 1.0e3).cpu().numpy()
            ctx.synchronize()
            if not torch.equal(out.view(it), x.view(it)):
                fails += 1
                print(f"FAIL {tag}: decoded bytes differ from the input")
            if dv[list(DEC_OPTS).index(capi.OPT_CONSUMER_PIPELINED)] != 1 and not same_or_nan(sums, ref_sums):  # (the ring kernel documents another order)
                fails += 1
                print(f"FAIL {tag}: sums differ")
            if not np.array_equal(cnt, ref_cnt):
                fails += 1
                print(f"FAIL {tag}: counts differ")
            del out
    except Exception as exc:  # noqa: BLE001
        fails += 1
        print(f"FAIL {what}: {type(exc).__name__}: {exc}")
    finally:
        defaults(ENC_OPTS), defaults(DEC_OPTS)
    done += 1
    del x, col, ref
print(f"fuzz_options: {done} rounds, {fails} failures, {time.time() - t0:.0f} s, lib {capi.LIB_PATH}", flush=True)
sys.exit(1 if fails else 0)
