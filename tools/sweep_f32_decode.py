#!/usr/bin/env python3
"""sweep_f32_decode.py [n_vectors]: the float store decode by packed width (WIDTHS, default 1..32), without and with 20 exceptions per vector: vectors per
workgroup 1 / 2 / 4 cold, and with the read-ahead beside (ALPGPU_OPT_DECODE_READ_AHEAD = 1) at a few leads — the data behind decode_policy.hpp's float
limits (round 6).  Fractions of 8 TB/s over algorithmic bytes; profiles/r06_float_decode.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from alp_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
widths = [int(w) for w in os.environ.get("WIDTHS", ",".join(str(w) for w in range(1, 33))).split(",")]
excs = [int(e) for e in os.environ.get("EXCS", "0,20").split(",")]
leads = [int(x) for x in os.environ.get("LEADS", "0,20,40").split(",")]
ahead_upto = int(os.environ.get("AHEAD_UPTO", "12"))
ctx = capi.Context(0)
out = torch.empty(n * 1024, dtype=torch.float32, device="cuda:0")
print(f"lib {bench.lib_sha16()}  n={n}")
print("bw exc |  vpw1   vpw2   vpw4   one wave per vector | auto(vpw,ahead) | read-ahead on, by (vpw, lead us): ...")
for exc in excs:
    for bw in widths:
        c, _, ab = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=bw, exc_per_vec=exc, value_bytes=4)
        f = lambda ms: ab / ms / 1e6 / 8000  # noqa: E731
        row = []
        ctx.set_option(capi.OPT_DECODE_READ_AHEAD, 0)
        for vpw in (1, 2, 4, 8):
            ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
            row.append(f(bench.time_launches(lambda: ctx.decode(c, out), 7, 4)[0]))
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)
        ctx.set_option(capi.OPT_DECODE_READ_AHEAD, -1)
        auto = f(bench.time_launches(lambda: ctx.decode(c, out), 7, 4)[0])
        shape = (ctx.decode_vectors_per_wg(c), int(ctx.decode_reads_ahead(c)))
        ra = []
        if bw <= ahead_upto:
            ctx.set_option(capi.OPT_DECODE_READ_AHEAD, 1)
            for vpw in (2, 8):
                for lead in leads:
                    ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
                    ctx.set_option(capi.OPT_DECODE_READ_AHEAD_US, lead)
                    ra.append(f"({vpw},{lead}):{f(bench.time_launches(lambda: ctx.decode(c, out), 7, 4)[0]):.3f}")
            ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)
            ctx.set_option(capi.OPT_DECODE_READ_AHEAD_US, 0)
            ctx.set_option(capi.OPT_DECODE_READ_AHEAD, -1)
        print(f"{bw:2d} {exc:3d} | {row[0]:.3f}  {row[1]:.3f}  {row[2]:.3f}  {row[3]:.3f} | {auto:.3f} {shape} | {' '.join(ra)}", flush=True)
        del c
