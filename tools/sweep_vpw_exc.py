#!/usr/bin/env python3
"""decode: one against two vectors per workgroup over (bit width x exceptions per vector), 1 Mi vectors, plus GPU-encoded mixed / ALP_RD
columns: the data behind the launch-shape rule of alpgpu_decode_f64 (api.hip: decode_variant_for).  sweep_vpw_exc.py [n_vectors]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench
from alp_amd import capi
from bench_decode_variants import make_column, timeit
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
ctx = capi.Context(0)
out = torch.empty(n * 1024, dtype=torch.float64, device="cuda")
def both(col):
    r = []
    for v in (1, 2):
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, v)
        ms, _ = timeit(lambda: ctx.decode(col, out), iters=9, warmup=3)
        r.append(ms)
    ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)
    return r
for exc in (0, 2, 5, 20, 100):
    line = []
    for bw in (8, 12, 16, 18, 20, 24, 28, 32, 40, 48):
        col, rec = make_column(n, bw, exc, seed=bw)
        a, b = both(col)
        alg = n * (32 + 128 * bw + rec + 8192)
        line.append(f"bw{bw}: {alg/a/8e9:.3f}/{alg/b/8e9:.3f}{'*' if b < a else ' '}")
        del col
        torch.cuda.empty_cache()
    print(f"exc={exc:3d}  (V=1/V=2, * = two wins)  " + "  ".join(line), flush=True)
for kind in ("mixed", "rd"):
    x = bench.synthetic_input(kind, n, torch.device("cuda:0"), seed=42)
    col = ctx.encode(x)
    ctx.synchronize()
    pb, eb, ov = ctx.column_totals(col)
    a, b = both(col)
    alg = n * (32 + 8192) + pb + eb
    print(f"{kind}: packed {pb/n:.0f} B/vec exc {eb/n:.0f} B/vec  V=1 {alg/a/8e9:.3f}  V=2 {alg/b/8e9:.3f}", flush=True)
    del x, col
