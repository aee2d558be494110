#!/usr/bin/env python3
"""sweep_pad_narrow.py: double columns of narrow vectors (BWS, default 2,3,4,6,8; EXCS 0,20), the store decode with the read-ahead beside it, by vectors per workgroup and
residency pad (KiB of unused LDS per workgroup: fewer resident workgroups = fewer bytes of stores in flight per CU).  Fractions of 8 TB/s."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from alp_amd import capi  # noqa: E402

n = 1 << 20
bws = [int(w) for w in os.environ.get("BWS", "2,3,4,6,8").split(",")]
excs = [int(e) for e in os.environ.get("EXCS", "0,20").split(",")]
pads = [int(e) for e in os.environ.get("PADS", "0,6,11,16,24,32").split(",")]
ctx = capi.Context(0)
out = torch.empty(n * 1024, dtype=torch.float64, device="cuda:0")
print(f"lib {bench.lib_sha16()}")
for exc in excs:
    for bw in bws:
        c, _, ab = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=bw, exc_per_vec=exc)
        f = lambda ms: ab / ms / 1e6 / 8000  # noqa: E731
        auto = f(bench.time_launches(lambda: ctx.decode(c, out), 7, 4)[0])
        row = []
        for ahead in (1, 0):
            ctx.set_option(capi.OPT_DECODE_READ_AHEAD, ahead)
            for vpw in (1, 2):
                ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
                for pad in pads:
                    ctx.set_option(capi.OPT_DECODE_RESIDENCY_PAD, pad)
                    row.append(f"a{ahead}v{vpw}p{pad}:{f(bench.time_launches(lambda: ctx.decode(c, out), 7, 4)[0]):.3f}")
        ctx.set_option(capi.OPT_DECODE_RESIDENCY_PAD, -1)
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)
        ctx.set_option(capi.OPT_DECODE_READ_AHEAD, -1)
        print(f"bw {bw:2d} exc {exc:3d} | auto {auto:.3f} (vpw {ctx.decode_vectors_per_wg(c)}, ahead {int(ctx.decode_reads_ahead(c))}) | " + " ".join(row), flush=True)
        del c
