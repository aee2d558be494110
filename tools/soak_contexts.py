#!/usr/bin/env python3
"""soak_contexts.py: 400 contexts created, used (encode, decode, SUM, totals) and destroyed, then 3 000 encode / decode calls over columns of random lengths in one
context; prints the free HBM against the start (a leak would grow: call 57 — 58 MiB after the first context and flat; 390 MiB of torch's cache, flat)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, datagen
from alp_amd import capi
x = torch.from_numpy(np.concatenate([datagen.mixed_column(3000, seed=1), datagen.rd_column(500, seed=2)])).cuda()
def free(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0]
c0 = capi.Context(0); col = c0.encode(x); out = c0.decode(col); c0.synchronize(); del c0
base = free()
for i in range(400):
    c = capi.Context(0)
    c.encode(x, col); c.decode(col, out); c.decode_sum(col); c.column_totals(col); c.decode(col, out)
    c.synchronize(); c.close(); del c
    if i % 100 == 99: print(i + 1, "contexts: free HBM delta", (base - free()) / 2**20, "MiB", flush=True)
c = capi.Context(0)
for i in range(3000):
    n = int(np.random.default_rng(i).integers(1, 3500))
    xi = x[: n * 1024]
    ci = c.encode(xi); o = c.decode(ci); c.column_sum(ci) if hasattr(c, "column_sum") else None
    if i % 1000 == 999: print(i + 1, "calls: free HBM delta", (base - free()) / 2**20, "MiB", flush=True)
print("RESULT ok" if torch.equal(o.view(torch.int64), xi.view(torch.int64)) else "RESULT differ")
