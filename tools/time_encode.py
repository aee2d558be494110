#!/usr/bin/env python3
"""time_encode.py [n_vectors] [kinds...]: alpgpu_encode_f64 (rowgroup search + vector encode) on bench.py's encode columns with the search
BESIDE the vector encode (ALPGPU_OPT_ENCODE_ASYNC_INIT = 1, the default) and in front of it (0); the two parts on their own; and a
byte comparison of everything the two routes write.  One process = one library (ALPGPU_LIB selects an A/B build)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from alp_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
kinds = sys.argv[2:] or ["mixed", "rd"]
ctx = capi.Context(0)
tag = os.path.basename(os.environ.get("ALPGPU_LIB", "libalpgpu.so")) + " kernel=" + os.environ.get("ALPGPU_ENCODE_KERNEL", "0(lean)") + (" wg/cu=" + os.environ["ALPGPU_ASYNC_INIT_WG_PER_CU"] if os.environ.get("ALPGPU_ASYNC_INIT_WG_PER_CU") else "")
for kind in kinds:
    x = bench.synthetic_input(kind, n, torch.device("cuda:0"), seed=42)
    cols, ms = {}, {}
    for mode in (0, 1, 0, 1):
        ctx.set_option(capi.OPT_ENCODE_ASYNC_INIT, mode)
        col = capi.DeviceColumn(n, 0)
        ms[mode], _ = bench.time_launches(lambda: ctx.encode(x, col), 7, 3)
        cols[mode] = col
    ctx.set_option(capi.OPT_ENCODE_ASYNC_INIT, 1)
    imed, _ = bench.time_launches(lambda: ctx.rowgroup_init(x, cols[0]), 5, 2)
    vmed, _ = bench.time_launches(lambda: ctx.encode_vectors(x, cols[0]), 5, 2)
    pb, eb, ov = ctx.column_totals(cols[1])
    same = all(torch.equal(a, b) for a, b in ((cols[0].rowgroups, cols[1].rowgroups), (cols[0].vectors, cols[1].vectors), (cols[0].packed[:pb], cols[1].packed[:pb]), (cols[0].exc[:eb], cols[1].exc[:eb])))
    alg = bench.encode_alg_bytes(n, pb, eb)
    print(f"{tag} {kind} n={n}: search beside the encode {ms[1]:.3f} ms = {alg / ms[1] / 1e6 / 8000:.3f} of peak | search in front {ms[0]:.3f} ms = {alg / ms[0] / 1e6 / 8000:.3f} "
          f"| search alone {imed:.3f} | vectors alone {vmed:.3f} | the two routes wrote the same bytes: {same}", flush=True)
    del x, cols
