#!/usr/bin/env python3
"""time_consumers.py [n_vectors]: the fused consumers on bench.py's configs[1] column (bit widths 1-53 by rowgroup) and on columns with
exceptions — alpgpu_decode_sum_f64 (default: one workgroup per two vectors), the count consumer, the column total, and the SUM
through the persistent LDS-ring kernel (ALPGPU_OPT_CONSUMER_PIPELINED).  One process = one library (ALPGPU_LIB selects an A/B build: tools/build_variant.sh)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from alp_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
ctx = capi.Context(0)
tag = os.path.basename(os.environ.get("ALPGPU_LIB", "libalpgpu.so"))
sums = torch.empty(n, dtype=torch.float64, device="cuda")
cnts = torch.empty(n, dtype=torch.int32, device="cuda")
for label, kw in (("bw1-53", {}), ("bw4", {"bw_of_rowgroup": 4}), ("bw16", {"bw_of_rowgroup": 16}), ("bw16_exc20", {"bw_of_rowgroup": 16, "exc_per_vec": 20}),
                  ("bw48", {"bw_of_rowgroup": 48}), ("mixed (GPU-encoded, ~100 exceptions per vector)", "mixed"), ("rd (GPU-encoded ALP_RD)", "rd")):
    if isinstance(kw, str):
        x = bench.synthetic_input(kw, n, torch.device("cuda:0"), seed=42)
        col = ctx.encode(x)
        ctx.synchronize()
        pb, eb, ov = ctx.column_totals(col)
        del x
        read = pb + eb + 13 * n + 8 * n
    else:
        col, vec, alg = bench.build_decode_column(n, 0, seed=42, **kw)
        read = alg - n * 8192 + n * 8
    row = []
    for name, fn in (("sum", lambda: ctx.decode_sum(col, sums)), ("count", lambda: ctx.decode_count_range(col, -1.0, 1.0, cnts)), ("column_sum", lambda: ctx.column_sum(col))):
        med, _ = bench.time_launches(fn, 7, 5)
        row.append(f"{name} {med:.3f} ms = {read / med / 1e6 / 8000:.3f} of peak")
    ctx.set_option(capi.OPT_CONSUMER_PIPELINED, 1)
    med, _ = bench.time_launches(lambda: ctx.decode_sum(col, sums), 7, 5)
    ctx.set_option(capi.OPT_CONSUMER_PIPELINED, 0)
    row.append(f"pipelined kernel sum {med:.3f} ms = {read / med / 1e6 / 8000:.3f}")
    ctx.set_option(capi.OPT_CONSUMER_PIPELINED, 2)
    med, _ = bench.time_launches(lambda: ctx.decode_sum(col, sums), 7, 5)
    cmed, _ = bench.time_launches(lambda: ctx.decode_count_range(col, -1.0, 1.0, cnts), 7, 5)
    ctx.set_option(capi.OPT_CONSUMER_PIPELINED, 0)
    row.append(f"one wavefront per vector: sum {med:.3f} ms = {read / med / 1e6 / 8000:.3f}, count {cmed:.3f}")
    print(f"{tag} {label} n={n}: " + " | ".join(row), flush=True)
    del col
