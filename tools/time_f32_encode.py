#!/usr/bin/env python3
"""time_f32_encode.py [n]: the float encode (rowgroup search + vectors) of bench.py's two float columns — the search in front (default) and beside
(ALPGPU_OPT_ENCODE_ASYNC_INIT 2), ordered and unordered; ms per launch and fractions of 8 TB/s over algorithmic bytes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from alp_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
VEC = 1024
dev = "cuda:0"
ctx = capi.Context(0)
print(f"lib {bench.lib_sha16()}")
for kind in ("decimal_mixed", "rd"):
    g = torch.Generator(device=dev)
    g.manual_seed(43)
    if kind == "rd":
        xf = torch.rand(n * VEC, dtype=torch.float32, device=dev, generator=g)
    else:
        xd = (torch.rand(n * VEC, dtype=torch.float64, device=dev, generator=g) - 0.5) * 2e3
        sc = torch.where((torch.arange((n + 99) // 100, device=dev) % 2 == 0), 10.0, 100.0).to(torch.float64).repeat_interleave(100 * VEC)[: n * VEC]
        xf = (torch.round(xd * sc) / sc).to(torch.float32)
        m = torch.rand(n * VEC, device=dev, generator=g) < 0.01
        xf[m] = (xd[m] * 3.141592653589793).to(torch.float32)
        del xd, sc, m
    fcol = capi.DeviceColumn(n, 0, dtype="f32")
    ctx.encode(xf, fcol)
    pb, eb, ov = ctx.column_totals(fcol)
    ref = [t.clone() for t in (fcol.packed[:pb], fcol.exc[:eb])]
    f_alg = n * (4096 + 13) + pb + eb
    row = []
    for unordered in (0, 1):
        for async_init in (1, 2):
            ctx.set_option(capi.OPT_ENCODE_UNORDERED, unordered)
            ctx.set_option(capi.OPT_ENCODE_ASYNC_INIT, async_init)
            med, _ = bench.time_launches(lambda: ctx.encode(xf, fcol), 5, 2)
            same = ""
            if not unordered:
                ctx.synchronize()
                same = " same bytes" if torch.equal(fcol.packed[:pb], ref[0]) and torch.equal(fcol.exc[:eb], ref[1]) else " BYTES DIFFER"
            row.append(f"{'unordered' if unordered else 'ordered'}, search {'beside' if async_init == 2 else 'in front'}: {med:.3f} ms = {f_alg / med / 1e6 / 8000:.3f}{same}")
    ctx.set_option(capi.OPT_ENCODE_UNORDERED, 0)
    ctx.set_option(capi.OPT_ENCODE_ASYNC_INIT, 1)
    print(f"{kind:14s} " + " | ".join(row), flush=True)
    del xf, fcol
