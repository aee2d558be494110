#!/usr/bin/env python3
"""r05_read_ahead_pads.py: narrow columns under the read-ahead: residency pad x lead, one vector per workgroup (and two, pad 0).  Env: WIDTHS, EXCS, LEADS, PADS."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from alp_amd import capi  # noqa: E402

n = 1 << 20
widths = [int(w) for w in os.environ.get("WIDTHS", "1,2,3,4,6,8").split(",")]
excs = [int(e) for e in os.environ.get("EXCS", "0,20").split(",")]
leads = [int(v) for v in os.environ.get("LEADS", "20,40,60").split(",")]
pads = [int(v) for v in os.environ.get("PADS", "0,6,11,14,20").split(",")]
ctx = capi.Context(0)
out = torch.empty(n * 1024, dtype=torch.float64, device="cuda:0")
print(f"lib {bench.lib_sha16()} grid {os.environ.get('ALPGPU_READ_AHEAD_GRID', 'default')}: bw exc | plain auto | per lead {leads} us: [vpw1 at pads {pads} ; vpw2 pad 0]", flush=True)
for exc in excs:
    for w in widths:
        c, _, ab = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=w, exc_per_vec=exc)

        def frac():
            med, _ = bench.time_launches(lambda: ctx.decode(c, out), 7, 3)
            return ab / med / 1e6 / 8000

        ctx.set_option(capi.OPT_DECODE_READ_AHEAD, 0)
        ctx.set_option(capi.OPT_DECODE_RESIDENCY_PAD, -1)
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)
        plain = frac()
        ctx.set_option(capi.OPT_DECODE_READ_AHEAD, 1)
        line = f"{w:>3} {exc:>3} | {plain:.3f} |"
        for us in leads:
            ctx.set_option(capi.OPT_DECODE_READ_AHEAD_US, us)
            ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 1)
            r = []
            for pad in pads:
                ctx.set_option(capi.OPT_DECODE_RESIDENCY_PAD, pad)
                r.append(frac())
            ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 2)
            ctx.set_option(capi.OPT_DECODE_RESIDENCY_PAD, 0)
            two = frac()
            line += " [" + " ".join(f"{f:.3f}" for f in r) + f" ; {two:.3f}]"
        print(line, flush=True)
        del c
