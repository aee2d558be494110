#!/usr/bin/env python3
"""sweep_sink_residency.py: alpgpu_decode_sum_f64 (k_sink_direct) on the benchmark column and single widths under ALPGPU_SINK_PAD_LDS_KIB (workgroups
resident per CU); fraction of the HBM peak on the sink's read-only algorithmic bytes"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from alp_amd import capi
n = 1 << 20
ctx = capi.Context(0)
sums = torch.empty(n, dtype=torch.float64, device="cuda:0")
row = []
for bw, exc in ((None, 0), (4, 0), (8, 0), (16, 0), (24, 0), (32, 0), (40, 0), (53, 0), (None, 20), (16, 20), (40, 20)):
    c, _, ab = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=bw, exc_per_vec=exc)
    alg = ab - n * 8192 + n * 8
    best = 1e9
    for rnd in range(2):
        med, _ = bench.time_launches(lambda: ctx.decode_sum(c, sums), 7, 6)
        best = min(best, med)
    row.append(f"{'mix' if bw is None else bw}{'e' if exc else ''}:{alg / best / 1e6 / 8000:.3f}")
    del c
print(f"sink pad {os.environ.get('ALPGPU_SINK_PAD_LDS_KIB', '0'):>2s} KiB: " + "  ".join(row), flush=True)
