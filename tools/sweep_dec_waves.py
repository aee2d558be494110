#!/usr/bin/env python3
"""decode roofline fraction by bit width for 1 and 2 vectors per workgroup, for the library given by ALPGPU_LIB (e.g. an
-DALPGPU_DEC_WAVES=8 -DALPGPU_EXPERIMENT_DEC_WAVES build: 8 instead of 4 wavefronts per vector); checks the output bits"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alp_amd import capi
from bench import build_decode_column, time_launches, HBM_PEAK_GBPS
n = 1 << 20
ctx = capi.Context(0)
out = torch.empty(n * 1024, dtype=torch.float64, device="cuda:0")
tag = os.path.basename(os.environ.get("ALPGPU_LIB", "libalpgpu.so"))
for bw in [int(a) for a in os.environ.get("SWEEP_BWS", "1,2,4,8,16,27,40,53").split(",")]:
    c, _, ab = build_decode_column(n, 0, seed=7, bw_of_rowgroup=bw)
    row = []
    for vpw in (1, 2):
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
        med, _ = time_launches(lambda: ctx.decode(c, out), 7, 10)
        row.append(round(ab / med / 1e6 / HBM_PEAK_GBPS, 4))
    print(tag, "bw", bw, "V=1", row[0], "V=2", row[1], "checksum", int(out.view(torch.int64).sum().item()) & 0xFFFFFFFF)
    del c
