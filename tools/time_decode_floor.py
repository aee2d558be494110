#!/usr/bin/env python3
"""decode floor cases in ONE process: narrow bit widths, exception-heavy vectors, the mixed headline column; store kernel and SUM sink.
time_decode_floor.py [n_vectors]   (A/B: ALPGPU_LIB=build/variants/libalpgpu_<name>.so)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from alp_amd import capi
from bench_decode_variants import make_column, timeit
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 19
ctx = capi.Context(0)
out = torch.empty(n * 1024, dtype=torch.float64, device="cuda")
sums = torch.empty(n, dtype=torch.float64, device="cuda")
vpw = int(os.environ.get("VPW", "0"))  # hand-built columns carry no size hints: 0 = one vector per workgroup for all of them
ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
print("lib", os.environ.get("ALPGPU_LIB", "default"), "vectors per workgroup:", vpw or "auto (1 without hints)")
for bw, exc in [(1, 0), (2, 0), (3, 0), (4, 0), (8, 0), (16, 0), (28, 0), (48, 0), (16, 20), (28, 20), (28, 100), (16, 200)]:
    col, rec = make_column(n, bw, exc, seed=bw)
    ms, mn = timeit(lambda: ctx.decode(col, out), iters=9, warmup=3)
    ms2, mn2 = timeit(lambda: ctx.decode_sum(col, sums), iters=9, warmup=3)
    alg = n * (32 + 128 * bw + rec + 8192)
    rd = n * (32 + 128 * bw + rec + 8)
    print(f"bw={bw:2d} exc={exc:3d}: store {ms:.3f} ms frac {alg/ms/1e6/8e3:.3f} | sum {ms2:.3f} ms frac {rd/ms2/1e6/8e3:.3f}", flush=True)
    del col
    torch.cuda.empty_cache()
import datagen
host = np.concatenate([datagen.mixed_column(300, seed=1), datagen.rd_column(100, seed=2)])
reps = max(1, n // 400)
x = torch.from_numpy(host).cuda().repeat(reps)
col = ctx.encode(x)
ctx.synchronize()
nv = x.numel() // 1024
o2 = out[: x.numel()]
ms, mn = timeit(lambda: ctx.decode(col, o2), iters=9, warmup=3)
print(f"mixed+rd column {nv} vectors: store {ms:.3f} ms  {nv*8192/ms/1e6:.0f} GB/s out", flush=True)
assert torch.equal(o2.view(torch.int64), x.view(torch.int64))
