#!/usr/bin/env python3
"""prof_sink_direct_f32.py [n]: a few launches of both float SUM sinks (one wavefront per vector, then the staged one) on bench.py's float decimal column, for rocprofv3"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alp_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
dev = torch.device("cuda:0")
ctx = capi.Context(0)
g = torch.Generator(device=dev); g.manual_seed(43)
xd = (torch.rand(n * 1024, dtype=torch.float64, device=dev, generator=g) - 0.5) * 2e3
xf = (torch.round(xd * 10.0) / 10.0).to(torch.float32)
del xd
col = ctx.encode(xf)
ctx.synchronize()
print(ctx.column_totals(col))
sums = torch.empty(n, dtype=torch.float64, device=dev)
for mode in (2, 3):
    ctx.set_option(capi.OPT_CONSUMER_PIPELINED, mode)
    for _ in range(5):
        ctx.decode_sum(col, sums)
torch.cuda.synchronize()
