#!/usr/bin/env python3
"""prof_decode_narrow.py [n]: a few launches of the store decode on narrow columns — double 3 bits (one / two vectors per workgroup) and float 3 bits (two / four per
workgroup, one wavefront per vector), read-ahead off — for rocprofv3 counters (tools/pmc_busy.sh): what a float workgroup does more than a double one"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from alp_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
ctx = capi.Context(0)
ctx.set_option(capi.OPT_DECODE_READ_AHEAD, 0)
out = torch.empty(n * 1024, dtype=torch.float64, device="cuda:0")
for vb, shapes in ((8, (1, 2)), (4, (2, 4, 8))):
    for exc in (0, 20):
        c, _, ab = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=3, exc_per_vec=exc, value_bytes=vb)
        o = out if vb == 8 else out.view(torch.float32)[: n * 1024]
        for vpw in shapes:
            ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
            for _ in range(3):
                ctx.decode(c, o)
            torch.cuda.synchronize()
            print(f"vb {vb} exc {exc} vpw {vpw}: {ab / bench.time_launches(lambda: ctx.decode(c, o), 3, 1)[0] / 1e6 / 8000:.3f}")
        del c
