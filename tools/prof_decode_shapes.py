#!/usr/bin/env python3
"""prof_decode_shapes.py [n]: the decode under the auto rule on columns that take its other launch shapes (for rocprofv3): 8 bits with 20 exceptions per
vector (k_decode_pairs), 12 bits (two vectors per workgroup, six workgroups per CU), 44 bits (one vector per workgroup, six per CU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from alp_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
ctx = capi.Context(0)
out = torch.empty(n * 1024, dtype=torch.float64, device="cuda:0")
for bw, exc in ((8, 20), (12, 0), (44, 0)):
    c, _, ab = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=bw, exc_per_vec=exc)
    med, _ = bench.time_launches(lambda: ctx.decode(c, out), 7, 3)
    print(f"bw {bw} exceptions {exc}: {med:.3f} ms = {ab / med / 1e6 / 8000:.3f} of peak ({ctx.decode_vectors_per_wg(c)} vector(s) per workgroup)", flush=True)
    del c
