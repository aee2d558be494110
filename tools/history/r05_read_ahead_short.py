#!/usr/bin/env python3
"""r05_read_ahead_short.py: where the read-ahead starts to pay by column LENGTH: 3-bit and 6-bit columns of 32 Ki .. 1 Mi vectors, without / with 20 exceptions,
option off (0) against forced on (1), both shapes.  COLD=1: 2 GiB read between launches (a column decoded once is cold; back-to-back launches of a short column find
their inputs in the Infinity Cache).  Fractions of 8 TB/s."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from alp_amd import capi  # noqa: E402

ctx = capi.Context(0)
COLD = os.environ.get("COLD", "0") == "1"
flush = torch.ones(1 << 29, dtype=torch.float32, device="cuda:0") if COLD else None
out = torch.empty((1 << 20) * 1024, dtype=torch.float64, device="cuda:0")
print(f"lib {bench.lib_sha16()} cold {COLD}: n bw exc | off | on (vpw1 / vpw2)", flush=True)
for n in (32768, 65536, 131072, 262144, 524288, 1048576):
    for bw in (3, 6):
        for exc in (0, 20):
            c, _, ab = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=bw, exc_per_vec=exc)

            def frac():
                if not COLD:
                    med, _ = bench.time_launches(lambda: ctx.decode(c, out), 15, 5)
                    return ab / med / 1e6 / 8000
                ts = []
                for _ in range(9):
                    flush.sum()  # 2 GiB through the Infinity Cache: the column's streams are cold again
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    ctx.decode(c, out)
                    b.record()
                    torch.cuda.synchronize()
                    ts.append(a.elapsed_time(b))
                ts.sort()
                return ab / ts[len(ts) // 2] / 1e6 / 8000

            ctx.set_option(capi.OPT_DECODE_READ_AHEAD, 0)
            ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)
            off = frac()
            ctx.set_option(capi.OPT_DECODE_READ_AHEAD, 1)
            ctx.set_option(capi.OPT_DECODE_RESIDENCY_PAD, 0)
            on = []
            for vpw in (1, 2):
                ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
                on.append(frac())
            ctx.set_option(capi.OPT_DECODE_RESIDENCY_PAD, -1)
            print(f"{n:>8} {bw} {exc:>2} | {off:.3f} | {on[0]:.3f} / {on[1]:.3f}", flush=True)
            del c
