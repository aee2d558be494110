#!/usr/bin/env python3
"""sweep_crossover.py: one against two vectors per decode workgroup around the crossover (10..20 bits), three rounds each — boxes disagree by a few
percent there (profiles/r04_decode_floor.txt)."""
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
    for bw in range(10, 23):
        c, _, ab = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=bw, exc_per_vec=exc)
        best = {1: 0.0, 2: 0.0}
        for rnd in range(3):
            for vpw in (1, 2):
                ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
                med, _ = bench.time_launches(lambda: ctx.decode(c, out), 7, 6)
                best[vpw] = max(best[vpw], ab / med / 1e6 / 8000)
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)
        print(f"exc {exc:2d} bw {bw:2d}: one {best[1]:.3f}  two {best[2]:.3f}  -> {1 if best[1] >= best[2] else 2}  (auto picks {ctx.decode_vectors_per_wg(c)})", flush=True)
        del c
