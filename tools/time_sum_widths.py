#!/usr/bin/env python3
"""time_sum_widths.py: the double SUM sink (default kernel) on the benchmark column (widths 1-53 by rowgroup) and on columns of one width; ms per 1 Mi vectors.
For A/B libraries (ALPGPU_LIB=...); the sums of the two libraries can be compared through SUMS_OUT / SUMS_REF (a .pt file of the benchmark column's sums)."""
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
c, _, ab = bench.build_decode_column(n, 0, seed=42)
print(f"benchmark column: {bench.time_launches(lambda: ctx.decode_sum(c, sums), 9, 5)[0]:.3f} ms")
if os.environ.get("SUMS_OUT"):
    torch.save(sums.cpu(), os.environ["SUMS_OUT"])
if os.environ.get("SUMS_REF"):
    ref = torch.load(os.environ["SUMS_REF"])
    print("same bits as the reference library:", bool(torch.equal(ref.view(torch.int64), sums.cpu().view(torch.int64))))
del c
row = []
for bw in (2, 6, 12, 16, 24, 32, 33, 44):
    for exc in (0, 20):
        c, _, ab = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=bw, exc_per_vec=exc)
        row.append(f"bw{bw}/{exc}:{bench.time_launches(lambda: ctx.decode_sum(c, sums), 7, 5)[0]:.3f}")
        del c
print(" ".join(row))
