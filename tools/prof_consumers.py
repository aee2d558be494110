#!/usr/bin/env python3
"""prof_consumers.py <bw|0> [n]: a few launches of the fused SUM consumer on one bench column, for rocprofv3: the default (one wavefront per
vector, k_sink_direct), the staged four-wavefront kernel (k_decode_column<2, false, kSinkSum>), the persistent ring kernel (k_consume_column)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from alp_amd import capi
bw = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
ctx = capi.Context(0)
col, vec, alg = bench.build_decode_column(n, 0, seed=42, **({"bw_of_rowgroup": bw} if bw else {}))
sums = torch.empty(n, dtype=torch.float64, device="cuda")
for _ in range(5):
    ctx.decode_sum(col, sums)
for mode in (3, 1):
    ctx.set_option(capi.OPT_CONSUMER_PIPELINED, mode)
    for _ in range(5):
        ctx.decode_sum(col, sums)
torch.cuda.synchronize()
