#!/usr/bin/env python3
"""time_rd_f64.py [n]: a GPU-encoded ALP_RD double column (uniform doubles in [0, 1)) — store decode, SUM by each kernel; fractions of 8 TB/s over algorithmic bytes.
For A/B libraries (ALPGPU_LIB=...)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from alp_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
ctx = capi.Context(0)
print(f"lib {bench.lib_sha16()}")
for kind in ("rd", "mixed"):
    x = bench.synthetic_input(kind, n, torch.device("cuda:0"), seed=42)
    col = ctx.encode(x)
    ctx.synchronize()
    pb, eb, ov = ctx.column_totals(col)
    out = torch.empty_like(x)
    alg = n * (8192 + 13) + pb + eb
    fr = lambda b, ms: b / ms / 1e6 / 8000  # noqa: E731
    d, _ = bench.time_launches(lambda: ctx.decode(col, out), 7, 6)
    rt = bool(torch.equal(out.view(torch.int64), x.view(torch.int64)))
    sums = torch.empty(n, dtype=torch.float64, device="cuda")
    row = []
    for mode in (0, 1, 2):
        ctx.set_option(capi.OPT_CONSUMER_PIPELINED, mode)
        m, _ = bench.time_launches(lambda: ctx.decode_sum(col, sums), 7, 5)
        row.append(f"sum[{mode}] {m:.3f} ms = {fr(alg - n * 8192 + n * 8, m):.3f}")
    ctx.set_option(capi.OPT_CONSUMER_PIPELINED, 0)
    print(f"{kind:6s} dec {d:.3f} ms = {fr(alg, d):.3f} (rt {rt}) | " + " | ".join(row), flush=True)
    del x, col, out, sums
