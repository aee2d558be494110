#!/usr/bin/env python3
"""time_sum_exc.py: the double SUM sink on columns of one width with 0 / 20 / 100 exceptions per vector (ms per 1 Mi vectors), and on the GPU-encoded mixed column.
For A/B libraries (ALPGPU_LIB=...)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from alp_amd import capi  # noqa: E402

n = 1 << 20
ctx = capi.Context(0)
print(f"lib {bench.lib_sha16()}")
sums = torch.empty(n, dtype=torch.float64, device="cuda")
for bw in (6, 16, 28, 44):
    row = []
    for exc in (0, 20, 100):
        c, _, ab = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=bw, exc_per_vec=exc)
        row.append(f"{bench.time_launches(lambda: ctx.decode_sum(c, sums), 7, 5)[0]:.3f}")
        del c
    print(f"bw {bw:2d}: " + "  ".join(row), flush=True)
x = bench.synthetic_input("mixed", n, torch.device("cuda:0"), seed=42)
col = ctx.encode(x)
ctx.column_totals(col)
print(f"mixed: {bench.time_launches(lambda: ctx.decode_sum(col, sums), 7, 5)[0]:.3f}")
