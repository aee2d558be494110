#!/usr/bin/env python3
"""r05_segments.py: columns whose regions differ, decoded in the shape of their average against region by region (ALPGPU_OPT_DECODE_SEGMENTS, after alpgpu_column_totals):
the bench's bimodal column (6 bits + 20 exceptions | 44 bits), thirds (3 bits | 28 bits | 12 bits + 20 exceptions), a narrow head of 30 % before wide vectors, the
benchmark column (widths by rowgroup: one kind of segment — must stay one launch), eight alternating stripes.  Fractions of 8 TB/s."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from alp_amd import capi  # noqa: E402

n = 1 << 20
ctx = capi.Context(0)
out = torch.empty(n * 1024, dtype=torch.float64, device="cuda:0")
i = np.arange(n)
h = n // 2 // 100 * 100
t1, t2 = n // 3 // 100 * 100, 2 * n // 3 // 100 * 100
cases = {
    "bimodal": (np.where(i < h, 6, 44), np.where(i < h, 20, 0)),
    "thirds": (np.where(i < t1, 3, np.where(i < t2, 28, 12)), np.where(i >= t2, 20, 0)),
    "narrow head": (np.where(i < 3 * n // 10, 4, 36), 0),
    "benchmark": (None, 0),
    "stripes": (np.where((i // (n // 8)) % 2 == 0, 5, 40), 0),
}
print(f"lib {bench.lib_sha16()}: case | average (vpw) | by segments (runs) | segments, read-ahead off", flush=True)
for name, (bw, exc) in cases.items():
    c, _, ab = bench.build_decode_column(n, 0, seed=9, bw_of_rowgroup=bw, exc_per_vec=exc)

    def frac():
        med, _ = bench.time_launches(lambda: ctx.decode(c, out), 9, 4)
        return ab / med / 1e6 / 8000

    avg, vpw = frac(), ctx.decode_vectors_per_wg(c)
    ctx.column_totals(c)
    seg, runs = frac(), ctx.decode_runs(c)
    ctx.set_option(capi.OPT_DECODE_READ_AHEAD, 0)
    seg0 = frac()
    ctx.set_option(capi.OPT_DECODE_READ_AHEAD, -1)
    ctx.set_option(capi.OPT_DECODE_SEGMENTS, 0)
    avg2 = frac()  # the average's shape once more, behind the others (order effects)
    ctx.set_option(capi.OPT_DECODE_SEGMENTS, 1)
    print(f"{name:>12} | {avg:.3f} ({vpw}) | {seg:.3f} ({runs}) | {seg0:.3f} | average again {avg2:.3f}", flush=True)
    del c
