#!/usr/bin/env python3
"""decode roofline fraction by bit width for 1 and 2 vectors per workgroup (where should the automatic choice switch?)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alp_amd import capi
from bench import build_decode_column, time_launches, HBM_PEAK_GBPS
n = 1 << 20
ctx = capi.Context(0)
out = torch.empty(n * 1024, dtype=torch.float64, device="cuda:0")
for bw in (8, 12, 14, 16, 18, 20, 22, 24, 28):
    c, _, ab = build_decode_column(n, 0, seed=7, bw_of_rowgroup=bw)
    row = []
    for vpw in (1, 2, 0):
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
        med, _ = time_launches(lambda: ctx.decode(c, out), 7, 10)
        row.append(round(ab / med / 1e6 / HBM_PEAK_GBPS, 4))
    print(bw, "V=1", row[0], "V=2", row[1], "auto", row[2])
    del c
