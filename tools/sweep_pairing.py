#!/usr/bin/env python3
"""sweep_pairing.py: ALPGPU_OPT_DECODE_PAIRING (k_decode_pairs) against the column-level launch-shape rule — on the benchmark column (widths 1..53 by
rowgroup), on its narrow and wide halves as columns of their own, and on single widths (profiles/r04_decode_floor.txt, section 4)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from alp_amd import capi
n = 1 << 20
ctx = capi.Context(0)
out = torch.empty(n * 1024, dtype=torch.float64, device="cuda:0")
rg = np.arange(n, dtype=np.int64) // 100
cases = [("benchmark column 1..53", None, 0), ("narrow half 1..16 by rowgroup", 1 + rg % 16, 0), ("wide half 17..53 by rowgroup", 17 + rg % 37, 0),
         ("benchmark widths, 20 exceptions", 1 + rg % 53, 20)]
cases += [(f"bw {b}", b, e) for e in (0, 20) for b in (2, 4, 8, 12, 14, 16, 18, 20, 24, 32, 40, 53)]
for name, bw, exc in cases:
    c, _, ab = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=bw, exc_per_vec=exc)
    row = []
    for rounds in range(2):
        for mode in ("auto", 1, 2, "p1", "p2", "p3"):
            ctx.set_option(capi.OPT_DECODE_PAIRING, int(mode[1]) if isinstance(mode, str) and mode[0] == "p" else 0)
            ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, mode if isinstance(mode, int) else 0)
            med, _ = bench.time_launches(lambda: ctx.decode(c, out), 7, 6)
            row.append(ab / med / 1e6 / 8000)
    ctx.set_option(capi.OPT_DECODE_PAIRING, 0)
    ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)
    best = [max(row[i], row[i + 6]) for i in range(6)]
    print(f"{name:34s} exc {exc:2d}: auto {best[0]:.3f} | one {best[1]:.3f} | two {best[2]:.3f} | pairs, one after the other {best[3]:.3f} | pairs, staggered {best[4]:.3f} | three per two {best[5]:.3f}", flush=True)
    del c
