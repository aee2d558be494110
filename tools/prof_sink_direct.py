#!/usr/bin/env python3
"""prof_sink_direct.py <bw|0> [n]: a few launches of the one-wavefront-per-vector SUM sink (ALPGPU_OPT_CONSUMER_PIPELINED = 2) on one bench column, for rocprofv3"""
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
ctx.set_option(capi.OPT_CONSUMER_PIPELINED, 2)
for _ in range(6):
    ctx.decode_sum(col, sums)
torch.cuda.synchronize()
