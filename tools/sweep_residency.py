#!/usr/bin/env python3
"""sweep_residency.py <pad KiB>: decode (one vector per workgroup) of single-width columns and the benchmark column under the residency cap that
ALPGPU_DECODE_PAD_LDS_KIB (unused dynamic LDS per workgroup) sets; run once per cap, the environment variable is read at the first decode."""
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
pad = os.environ.get("ALPGPU_DECODE_PAD_LDS_KIB", "0")
row = []
widths = [None if w == "mix" else int(w) for w in sys.argv[1].split(",")] if len(sys.argv) > 1 else [None, 18, 24, 32, 36, 40, 44, 48, 53]
for bw in widths:
    c, _, ab = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=bw, exc_per_vec=int(os.environ.get("SWEEP_EXC", "0")))
    ctx.set_option(capi.OPT_DECODE_PAIRING, int(os.environ.get("SWEEP_PAIRING", "0")))
    ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, int(os.environ.get("SWEEP_VPW", "1")))
    best = 0.0
    for rnd in range(2):
        med, _ = bench.time_launches(lambda: ctx.decode(c, out), 7, 6)
        best = max(best, ab / med / 1e6 / 8000)
    row.append(f"{'mix' if bw is None else bw}:{best:.3f}")
    del c
idle = os.environ.get("ALPGPU_DECODE_IDLE_FROM", "0") + "/" + os.environ.get("ALPGPU_DECODE_IDLE_UNITS", "0")
print(f"idle {idle:>5s} pad {pad:>3s} KiB ({'8' if int(pad) < 11 else str(int(160 // (int(pad) + 9.7)))} workgroups per CU): " + "  ".join(row), flush=True)
