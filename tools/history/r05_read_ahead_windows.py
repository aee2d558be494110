#!/usr/bin/env python3
"""r05_read_ahead_windows.py: the read-ahead's window (ALPGPU_OPT_DECODE_READ_AHEAD_US) per width and launch shape, 1 Mi-vector columns; pad 0.
Fractions of 8 TB/s.  Env: WIDTHS, EXCS, WINDOWS."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from alp_amd import capi  # noqa: E402

n = 1 << 20
widths = os.environ.get("WIDTHS", "2,4,6,8,12,16,20,28,mix").split(",")
excs = [int(e) for e in os.environ.get("EXCS", "0,20").split(",")]
windows = [int(v) for v in os.environ.get("WINDOWS", "5,10,20,40,80,160").split(",")]
ctx = capi.Context(0)
out = torch.empty(n * 1024, dtype=torch.float64, device="cuda:0")
print(f"lib {bench.lib_sha16()} grid {os.environ.get('ALPGPU_READ_AHEAD_GRID', 'default')} mode {os.environ.get('ALPGPU_READ_AHEAD_MODE', '0')}: bw exc | plain auto (vpw) | "
      f"vpw1 at leads {windows} us | vpw2 at the same", flush=True)
for exc in excs:
    for w in widths:
        if w == "mix" and exc:
            continue
        c, _, ab = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=None if w == "mix" else int(w), exc_per_vec=exc)

        def frac():
            med, _ = bench.time_launches(lambda: ctx.decode(c, out), 7, 3)
            return ab / med / 1e6 / 8000

        ctx.set_option(capi.OPT_DECODE_READ_AHEAD, 0)
        ctx.set_option(capi.OPT_DECODE_RESIDENCY_PAD, -1)
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)
        plain = frac()
        auto_vpw = ctx.decode_vectors_per_wg(c)
        ctx.set_option(capi.OPT_DECODE_READ_AHEAD, 1)
        ctx.set_option(capi.OPT_DECODE_RESIDENCY_PAD, 0)
        rows = []
        for vpw in (1, 2):
            ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
            r = []
            for mib in windows:
                ctx.set_option(capi.OPT_DECODE_READ_AHEAD_US, mib)
                r.append(frac())
            rows.append(" ".join(f"{f:.3f}" for f in r))
        print(f"{w:>3} {exc:>3} | {plain:.3f} ({auto_vpw}) | " + " | ".join(rows), flush=True)
        del c
