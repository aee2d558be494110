#!/usr/bin/env python3
"""Does the position of a configuration inside a long run change its timing?  (clock / power state, cache residue)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alp_amd import capi
from bench import build_decode_column, time_launches, VEC
n = 1 << 20
ctx = capi.Context(0)
out = torch.empty(n * VEC, dtype=torch.float64, device="cuda")
order = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "8,1,8,2,1,16,8,4,1").split(",")]
for bw in order:
    c, _, ab = build_decode_column(n, 0, seed=7, bw_of_rowgroup=bw)
    res = []
    for vpw in (0, 1, 2):
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
        med, mean = time_launches(lambda: ctx.decode(c, out), 7, 2)
        res.append(f"vpw{vpw}: {n * 8192 / med / 1e6:7.1f} GB/s ({ab / med / 1e6 / 8000:.3f})")
    ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)
    print(f"bw {bw:2d}  " + "  ".join(res), flush=True)
    del c
