#!/usr/bin/env python3
"""r05_time_decode_exc.py [n_vectors] [exceptions per vector]: the decode of ALP columns WITH exceptions, one column per bit width — the mask route
(ALPGPU_OPT_DECODE_PATCH_AFTER = 0, round 4's kernel arm and launch rule) against the patch-after-the-stores arm (round 5) under each launch shape:
auto (the rule), one / two vectors per workgroup, the pair kernel.  Fractions of the 8 TB/s peak (algorithmic bytes).  profiles/r05_decode_exceptions.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from alp_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
exc = int(sys.argv[2]) if len(sys.argv) > 2 else 20
widths = [int(w) for w in os.environ.get("WIDTHS", "1,2,3,4,6,8,10,12,14,16,17,18,20,22,24,28,32,36,40,44,48,53").split(",")]
ctx = capi.Context(0)
out = torch.empty(n * 1024, dtype=torch.float64, device="cuda:0")
arms = [("mask/auto", 0, 0, 0), ("patch/auto", 64, 0, 0), ("patch/vpw1", 64, 1, 0), ("patch/vpw2", 64, 2, 0), ("patch/pairs", 64, 0, 1), ("mask/vpw2", 0, 2, 0), ("mask/pairs", 0, 0, 1)]
print(f"lib {bench.lib_sha16()}  n={n}  exceptions per vector={exc}")
print("bw   " + "  ".join(f"{a[0]:>11}" for a in arms) + "   auto-shape(mask) auto-shape(patch)")
rows = []
for bw in widths:
    col, _, ab = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=bw, exc_per_vec=exc)
    fr, shapes = [], []
    for name, patch, vpw, pairing in arms:
        ctx.set_option(capi.OPT_DECODE_PATCH_AFTER, patch)
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
        ctx.set_option(capi.OPT_DECODE_PAIRING, pairing)
        med, _ = bench.time_launches(lambda: ctx.decode(col, out), 7, 5)
        fr.append(ab / med / 1e6 / 8000)
        if name.endswith("auto"):
            shapes.append(ctx.decode_vectors_per_wg(col))
    ctx.set_option(capi.OPT_DECODE_PAIRING, 0)
    ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)
    ctx.set_option(capi.OPT_DECODE_PATCH_AFTER, 64)
    rows.append(fr)
    print(f"{bw:<4} " + "  ".join(f"{f:11.3f}" for f in fr) + f"   {shapes[0]:>8} {shapes[1]:>16}", flush=True)
    del col
a = np.array(rows)
print("min  " + "  ".join(f"{f:11.3f}" for f in a.min(0)))
print("mean " + "  ".join(f"{f:11.3f}" for f in a.mean(0)))
best = a[:, 2:5].max(1)
print("best fixed shape of the patch arm per width: " + " ".join(f"{w}:{['vpw1', 'vpw2', 'pairs'][int(i)]}" for w, i in zip(widths, a[:, 2:5].argmax(1))))
print(f"patch arm, best shape per width: min {best.min():.3f} mean {best.mean():.3f}")
