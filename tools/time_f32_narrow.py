#!/usr/bin/env python3
"""time_f32_narrow.py: the float store decode on narrow columns (BWS, default 1,2,3,6; EXCS 0,20) — cold (1 Mi vectors) and warm (128 Ki: records and
descriptors stay in the Infinity Cache), vectors per workgroup 2 / 4 / 8 (one wavefront per vector), read-ahead off, and the rule's own choice.
Fractions of 8 TB/s over algorithmic bytes.  For A/B libraries (ALPGPU_LIB=build/variants/...): profiles/r06_float_decode.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from alp_amd import capi  # noqa: E402

bws = [int(w) for w in os.environ.get("BWS", "1,2,3,6").split(",")]
excs = [int(e) for e in os.environ.get("EXCS", "0,20").split(",")]
shapes = [int(e) for e in os.environ.get("SHAPES", "2,4,8").split(",")]
leads = [int(e) for e in os.environ.get("LEADS", "").split(",") if e]
sizes = [int(e) for e in os.environ.get("SIZES", f"{1 << 20},{1 << 17}").split(",")]
ctx = capi.Context(0)
print(f"lib {bench.lib_sha16()}")
for n in sizes:
    out = torch.empty(n * 1024, dtype=torch.float32, device="cuda:0")
    for exc in excs:
        for bw in bws:
            c, _, ab = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=bw, exc_per_vec=exc, value_bytes=4)
            f = lambda ms: ab / ms / 1e6 / 8000  # noqa: E731
            row = []
            ctx.set_option(capi.OPT_DECODE_READ_AHEAD, 0)
            for vpw in shapes:
                ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
                row.append(f"v{vpw}:{f(bench.time_launches(lambda: ctx.decode(c, out), 9, 4)[0]):.3f}")
                for lead in leads:  # the read-ahead beside this shape, lead in microseconds
                    ctx.set_option(capi.OPT_DECODE_READ_AHEAD, 1)
                    ctx.set_option(capi.OPT_DECODE_READ_AHEAD_US, lead)
                    row.append(f"+a{lead}:{f(bench.time_launches(lambda: ctx.decode(c, out), 9, 4)[0]):.3f}")
                    ctx.set_option(capi.OPT_DECODE_READ_AHEAD_US, 0)
                    ctx.set_option(capi.OPT_DECODE_READ_AHEAD, 0)
            ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)
            ctx.set_option(capi.OPT_DECODE_READ_AHEAD, -1)
            auto = f(bench.time_launches(lambda: ctx.decode(c, out), 9, 4)[0])
            print(f"n {n:8d} bw {bw:2d} exc {exc:3d} | {' '.join(row)} | auto {auto:.3f} (vpw {ctx.decode_vectors_per_wg(c)}, ahead {int(ctx.decode_reads_ahead(c))})", flush=True)
            del c
    del out
