#!/usr/bin/env python3
"""sweep_vpw_fine.py: decode of a 1 Mi-vector column of ONE bit width, one against two vectors per workgroup, for every width 1..40, without and
with 20 exceptions per vector — where exactly the launch-shape rule of decode_variant_for (api.hip) should switch."""
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
    for bw in list(range(1, 25)) + [28, 32, 40, 53]:
        c, _, ab = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=bw, exc_per_vec=exc)
        row = []
        for vpw in (1, 2, 4):
            ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
            med, _ = bench.time_launches(lambda: ctx.decode(c, out), 7, 6)
            row.append(ab / med / 1e6 / 8000)
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)
        print(f"exc {exc:2d} bw {bw:2d}: one vector per workgroup {row[0]:.3f}  two {row[1]:.3f}  four (narrow stage) {row[2]:.3f}  -> {(1, 2, 4)[row.index(max(row))]}  (auto picks {ctx.decode_vectors_per_wg(c)})", flush=True)
        del c
