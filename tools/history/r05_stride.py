#!/usr/bin/env python3
"""r05_stride.py: does the ALIGNMENT of a vector's packed record matter?  16-bit vectors (2 KiB records) decode at 0.83, 12-15 and 17 bits at 0.73-0.75.
Columns of bw-bit vectors whose records lie at a 2 KiB stride (as 16-bit ones do) against the dense layout; read-ahead off; the rule's shape and both forced."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from alp_amd import capi  # noqa: E402

n = 1 << 20
ctx = capi.Context(0)
ctx.set_option(capi.OPT_DECODE_READ_AHEAD, 0)
out = torch.empty(n * 1024, dtype=torch.float64, device="cuda:0")
print(f"lib {bench.lib_sha16()}: bw | dense: auto vpw1 vpw2 | 2 KiB stride: auto vpw1 vpw2", flush=True)
for bw in (10, 12, 13, 14, 15, 16):
    res = []
    for stride in (0, 2048):
        c, vec, _ = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=16 if stride else bw, exc_per_vec=0)
        if stride:
            vec = vec.copy()
            vec["bw"] = bw
            vec["packed_off"] = np.arange(n, dtype=np.uint64) * np.uint64(stride)
            c.vectors.copy_(torch.from_numpy(vec.view(np.uint8).reshape(-1)).to(c.vectors.device))
            c.c.packed_bytes_hint = n * 128 * bw  # what the rule sees: the vectors' bits, not the holes
        ab = n * (128 * bw + 13 + 8192)
        row = []
        for vpw in (0, 1, 2):
            ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
            med, _ = bench.time_launches(lambda: ctx.decode(c, out), 7, 3)
            row.append(ab / med / 1e6 / 8000)
        res.append(" ".join(f"{f:.3f}" for f in row))
        del c
    print(f"{bw:>2} | {res[0]} | {res[1]}", flush=True)
