#!/usr/bin/env python3
"""large-column round trip check: where does decode(encode(x)) first differ from x?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from alp_amd import capi
kind, n = sys.argv[1], int(sys.argv[2])
ctx = capi.Context(0)
if kind == "f32":
    x = (torch.round(torch.rand(n * 1024, device="cuda", dtype=torch.float64) * 1e5) / 100).to(torch.float32)
    it = torch.int32
else:
    from bench import synthetic_input
    x = synthetic_input("mixed", n, torch.device("cuda:0"), seed=42)
    it = torch.int64
col = capi.DeviceColumn(n, dtype="f32" if kind == "f32" else "f64")
for rep in range(2):
    ctx.encode(x, col)
    ctx.synchronize()
    try:
        pb, eb, ov = ctx.column_totals(col)
    except Exception as e:
        print("totals:", e); pb = eb = -1
    out = ctx.decode(col)
    ctx.synchronize()
    bad = (out.view(it) != x.view(it)).view(n, 1024).any(dim=1)
    nb = int(bad.sum())
    print(f"rep {rep}: packed {pb} exc {eb}; vectors with mismatches: {nb}")
    if nb:
        idx = torch.nonzero(bad).flatten()[:8].cpu().numpy()
        print(" first bad vectors:", idx, " tiles:", idx // 4, " blocks:", idx // 256)
        vec = col.vectors.cpu().numpy().view(capi.VECTOR_DTYPE)
        for v in idx[:4]:
            print("  v", v, vec[v], " prev:", vec[v - 1]["packed_off"], vec[v - 1]["bw"], vec[v-1]["exc_off"], vec[v-1]["exc_cnt"])
