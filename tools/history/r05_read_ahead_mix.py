#!/usr/bin/env python3
"""r05_read_ahead_mix.py: columns that are NOT narrow under a read-ahead that touches every descriptor but only the records of vectors of at most
ALPGPU_READ_AHEAD_BITS bits (0: descriptors only): the benchmark column (widths 1..53 by rowgroup), 28- and 44-bit columns, the bimodal column
(6 bits + 20 exceptions, then 44 bits).  One process per (grid, bits): both come from the environment.  Option off against forced on, the rule's own shape."""
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
tag = f"grid {os.environ.get('ALPGPU_READ_AHEAD_GRID', '64')} bits {os.environ.get('ALPGPU_READ_AHEAD_BITS', '128')} lead {os.environ.get('ALPGPU_READ_AHEAD_US', '40')}"
half = n // 2 // 100 * 100
cases = [("mix", None, 0), ("28", 28, 0), ("44", 44, 0), ("16", 16, 0),
         ("bimodal", np.where(np.arange(n) < half, 6, 44), np.where(np.arange(n) < half, 20, 0))]
row = []
for name, bw, exc in cases:
    c, _, ab = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=bw, exc_per_vec=exc)

    def frac():
        med, _ = bench.time_launches(lambda: ctx.decode(c, out), 9, 4)
        return ab / med / 1e6 / 8000

    ctx.set_option(capi.OPT_DECODE_READ_AHEAD, 0)
    off = frac()
    ctx.set_option(capi.OPT_DECODE_READ_AHEAD, 1)
    on = frac()
    row.append(f"{name} {off:.3f} -> {on:.3f}")
    del c
print(f"{tag}: " + " | ".join(row), flush=True)
