#!/usr/bin/env python3
"""time_encode_levers.py [n_vectors]: the vector encode on a column that repeats the first 8192 vectors of bench.py's mixed (or rd) column.
With a library built with -DALPGPU_EXPERIMENT_ENC_WRAP_TRAFFIC (ALPGPU_LIB=...) every vector's bytes come from / go to the same few MiB, i.e.
the identical instruction stream without the HBM traffic: what a better traffic shape could buy at most (profiles/r03_encode_levers.txt)."""
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
for kind in ("mixed", "rd"):
    head = bench.synthetic_input(kind, 8192, torch.device("cuda:0"), seed=42)
    x = head.repeat(n // 8192)
    del head
    col = capi.DeviceColumn(n, 0)
    out = {}
    for mode in (0, 1):
        ctx.set_option(capi.OPT_ENCODE_ASYNC_INIT, mode)
        out[mode], _ = bench.time_launches(lambda: ctx.encode(x, col), 7, 3)
    ctx.set_option(capi.OPT_ENCODE_ASYNC_INIT, 1)
    ctx.rowgroup_init(x, col)
    vmed, _ = bench.time_launches(lambda: ctx.encode_vectors(x, col), 7, 3)
    imed, _ = bench.time_launches(lambda: ctx.rowgroup_init(x, col), 5, 2)
    pb, eb, ov = ctx.column_totals(col)
    alg = bench.encode_alg_bytes(n, pb, eb)
    print(f"{tag} {kind} x128: search beside {out[1]:.3f} ms ({alg / out[1] / 1e6 / 8000:.3f}) | in front {out[0]:.3f} ms | vectors alone {vmed:.3f} ms ({alg / vmed / 1e6 / 8000:.3f}) | search alone {imed:.3f} ms", flush=True)
    del x, col
