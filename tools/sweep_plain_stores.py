#!/usr/bin/env python3
"""sweep_plain_stores.py: non-temporal (default) against plain stores of the decoded doubles, under the auto launch rule, by bit width"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from alp_amd import capi
n = 1 << 20
ctx = capi.Context(0)
out = torch.empty(n * 1024, dtype=torch.float64, device="cuda:0")
for exc in (0, 20):
    rows = {0: [], 1: []}
    for bw in (None, 1, 2, 3, 4, 6, 8, 12, 16, 20, 24, 32, 40, 53):
        c, _, ab = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=bw, exc_per_vec=exc)
        for plain in (0, 1):
            ctx.set_option(capi.OPT_DECODE_PLAIN_STORES, plain)
            best = 0.0
            for rnd in range(2):
                med, _ = bench.time_launches(lambda: ctx.decode(c, out), 7, 6)
                best = max(best, ab / med / 1e6 / 8000)
            rows[plain].append(f"{'mix' if bw is None else bw}:{best:.3f}")
        ctx.set_option(capi.OPT_DECODE_PLAIN_STORES, 0)
        del c
    print(f"exc {exc:2d} non-temporal: " + "  ".join(rows[0]), flush=True)
    print(f"exc {exc:2d} plain       : " + "  ".join(rows[1]), flush=True)
