#!/usr/bin/env python3
"""r05_headline_pads.py: the benchmark column (configs[1]: widths 1..53 by rowgroup) and the bimodal column under each residency pad and launch shape."""
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
        "bimodal": bench.build_decode_column(n, 0, seed=9, bw_of_rowgroup=np.where(np.arange(n) < half, 6, 44), exc_per_vec=np.where(np.arange(n) < half, 20, 0))}
print(f"lib {bench.lib_sha16()}")
for name, (c, _, ab) in cols.items():
    for vpw in (1, 2):
        row = []
        for pad in (-1, 0, 3, 6, 11, 14):
            ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
            ctx.set_option(capi.OPT_DECODE_RESIDENCY_PAD, pad)
            ts = [bench.time_launches(lambda: ctx.decode(c, out), 9, 6)[0] for _ in range(2)]
            row.append(f"pad {pad}: {ab / min(ts) / 1e6 / 8000:.4f}")
        print(f"{name} vpw{vpw}: " + "  ".join(row), flush=True)
    ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)
    ctx.set_option(capi.OPT_DECODE_RESIDENCY_PAD, -1)
    ts = [bench.time_launches(lambda: ctx.decode(c, out), 9, 6)[0] for _ in range(2)]
    print(f"{name} auto: {ab / min(ts) / 1e6 / 8000:.4f} (vpw {ctx.decode_vectors_per_wg(c)})", flush=True)
