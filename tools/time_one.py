#!/usr/bin/env python3
"""time the decode kernel at given (bw, n) combos in ONE process: time_one.py bw:n[:vectors_per_wg] ...   (hand-built columns carry no size
hints: without the third field the decode takes one vector per workgroup)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from alp_amd import capi
from bench_decode_variants import make_column, timeit
ctx = capi.Context(0)
for spec in sys.argv[1:]:
    parts = [int(x) for x in spec.split(":")]
    bw, n = parts[0], parts[1]
    ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, parts[2] if len(parts) > 2 else 0)
    col, rec = make_column(n, bw, 0, seed=bw)
    out = torch.empty(n * 1024, dtype=torch.float64, device="cuda")
    ms, mn = timeit(lambda: ctx.decode(col, out), iters=7, warmup=2)
    alg = n * (32 + 128 * bw + 8192)
    print(f"bw={bw:2d} n={n:8d}: median {ms:.3f} ms (min {mn:.3f})  {n*8192/ms/1e9:.2f} TB/s out  {alg/ms/1e9:.2f} TB/s traffic  packed_ptr%2MiB={col.packed.data_ptr() % (1<<21)} out_ptr%2MiB={out.data_ptr() % (1<<21)}", flush=True)
    del col, out
    torch.cuda.empty_cache()
