#!/usr/bin/env python3
"""r05_sum_exc.py: the double SUM sink (k_sink_direct) on columns with 0 / 20 / 100 exceptions per vector at a few widths — the route with exceptions spills 96 bytes
per lane (ISA: scratch stores behind the exception values' LDS reads); what does it cost?  Fractions of 8 TB/s on the bytes read."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from alp_amd import capi  # noqa: E402

n = 1 << 20
ctx = capi.Context(0)
sums = torch.empty(n, dtype=torch.float64, device="cuda:0")
tag = os.path.basename(os.environ.get("ALPGPU_LIB", "libalpgpu.so"))
print(f"{tag} {bench.lib_sha16()}: bw | sum frac (ms) at 0 / 20 / 100 exceptions per vector", flush=True)
for bw in (6, 16, 28, 44):
    row = []
    for exc in (0, 20, 100):
        c, _, ab = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=bw, exc_per_vec=exc)
        med, _ = bench.time_launches(lambda: ctx.decode_sum(c, sums), 9, 4)
        rb = ab - n * 8192 + n * 8
        row.append(f"{rb / med / 1e6 / 8000:.3f} ({med:.3f})")
        del c
    print(f"{bw:>2} | " + " | ".join(row), flush=True)
