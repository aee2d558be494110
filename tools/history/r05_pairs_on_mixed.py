#!/usr/bin/env python3
"""r05_pairs_on_mixed.py: k_decode_pairs (workgroups that own two vectors and decide from the two descriptors) against the column-level rule on columns that
MIX widths: the benchmark column (1..53 by rowgroup), the bimodal column, and a GPU-encoded mixed column."""
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
half = n // 2 // 100 * 100
cols = {"benchmark": bench.build_decode_column(n, 0, seed=42),
        "bimodal": bench.build_decode_column(n, 0, seed=9, bw_of_rowgroup=np.where(np.arange(n) < half, 6, 44), exc_per_vec=np.where(np.arange(n) < half, 20, 0)),
        "narrow_third": bench.build_decode_column(n, 0, seed=10, bw_of_rowgroup=np.where((np.arange(n) // 100) % 3 == 0, 5, 40))}
print(f"lib {bench.lib_sha16()}")
for name, (c, _, ab) in cols.items():
    row = []
    for label, vpw, pairing in (("auto", 0, 0), ("vpw1", 1, 0), ("vpw2", 2, 0), ("pairs1", 0, 1), ("pairs2", 0, 2), ("pairs3", 0, 3)):
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
        ctx.set_option(capi.OPT_DECODE_PAIRING, pairing)
        ts = [bench.time_launches(lambda: ctx.decode(c, out), 9, 6)[0] for _ in range(2)]
        row.append(f"{label} {ab / min(ts) / 1e6 / 8000:.4f}")
    ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)
    ctx.set_option(capi.OPT_DECODE_PAIRING, 0)
    print(f"{name}: " + "  ".join(row), flush=True)
