#!/usr/bin/env python3
"""sweep_residency_rule.py: the decode under the auto rule (launch shape AND residency from the column's size hints) on single widths 28..53 with and
without exceptions; run it once with ALPGPU_DECODE_PAD_LDS_KIB=0 (no residency cap) and once without the variable (the rule)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from alp_amd import capi
n = 1 << 20
ctx = capi.Context(0)
out = torch.empty(n * 1024, dtype=torch.float64, device="cuda:0")
tag = "no cap" if os.environ.get("ALPGPU_DECODE_PAD_LDS_KIB") == "0" else "rule  "
for exc in (0, 20):
    row = []
    for bw in (28, 32, 33, 34, 35, 36, 38, 39, 40, 44, 48, 53):
        c, _, ab = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=bw, exc_per_vec=exc)
        best = 0.0
        for rnd in range(2):
            med, _ = bench.time_launches(lambda: ctx.decode(c, out), 7, 6)
            best = max(best, ab / med / 1e6 / 8000)
        row.append(f"{bw}:{best:.3f}")
        del c
    print(f"{tag} exc {exc:2d}: " + "  ".join(row), flush=True)
