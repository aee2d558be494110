#!/usr/bin/env python3
"""time_decode_lib.py: the store decode of the benchmark column (widths 1..53 by rowgroup) and of single-width columns, fraction of 8 TB/s over
algorithmic bytes, median of 9 launches, with a checksum of the decoded bytes (A/B libraries must print the same one).  ALPGPU_LIB=... selects
the library; BWS=8,16,... the single widths; EXC=n exceptions per vector."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from alp_amd import capi  # noqa: E402

n = int(os.environ.get("N", 1 << 20))
exc = int(os.environ.get("EXC", "0"))
bws = [int(b) for b in os.environ.get("BWS", "12,20,28,36,48").split(",") if b]
ctx = capi.Context(0)
out = torch.empty(n * 1024, dtype=torch.float64, device="cuda")
row = [f"lib {bench.lib_sha16()} exc {exc}"]
for bw in [None] + bws:
    c, _, ab = bench.build_decode_column(n, 0, seed=42 if bw is None else 7, bw_of_rowgroup=bw, exc_per_vec=exc)
    ms = bench.time_launches(lambda: ctx.decode(c, out), 9, 5)[0]
    chk = int(out.view(torch.int64)[:: 4099].sum().item()) & 0xFFFFFFFF
    row.append(f"{'col' if bw is None else bw}: {ab / ms / 8e9:.3f} ({ms:.3f} ms, {chk:08x})")
    del c
print(" | ".join(row), flush=True)
