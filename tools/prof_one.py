#!/usr/bin/env python3
"""run the decode kernel a few times at one bit width (for rocprofv3 --pmc runs): prof_one.py <bw> [exc] [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from alp_amd import capi
from bench_decode_variants import make_column
bw = int(sys.argv[1]); exc = int(sys.argv[2]) if len(sys.argv) > 2 else 0; n = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 18
ctx = capi.Context(0)
col, rec = make_column(n, bw, exc, seed=bw)
out = torch.empty(n * 1024, dtype=torch.float64, device="cuda")
for _ in range(4):
    ctx.decode(col, out)
torch.cuda.synchronize()
