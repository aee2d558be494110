#!/usr/bin/env python3
"""time_read_ahead.py [n_vectors]: the double store decode of narrow columns with the read-ahead at the default lead and at leads that are too short (LEADS, us), for the
self-stretching lead of round 6 (ALPGPU_READ_AHEAD_ADAPT=0 in the environment: the lead stays as given).  One process per arm.  Fractions of 8 TB/s."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from alp_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
leads = [int(x) for x in os.environ.get("LEADS", "0,10,20,30").split(",")]
ctx = capi.Context(0)
out = torch.empty(n * 1024, dtype=torch.float64, device="cuda:0")
print(f"lib {bench.lib_sha16()}  n={n}  adapt={os.environ.get('ALPGPU_READ_AHEAD_ADAPT', '1')}")
for bw, exc in ((2, 0), (4, 0), (6, 0), (3, 20), (5, 20), (7, 20)):
    c, _, ab = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=bw, exc_per_vec=exc)
    f = lambda ms: ab / ms / 1e6 / 8000  # noqa: E731
    ctx.set_option(capi.OPT_DECODE_READ_AHEAD, 0)
    row = [f"off {f(bench.time_launches(lambda: ctx.decode(c, out), 7, 4)[0]):.3f}"]
    ctx.set_option(capi.OPT_DECODE_READ_AHEAD, -1)
    for lead in leads:
        ctx.set_option(capi.OPT_DECODE_READ_AHEAD_US, lead)
        row.append(f"lead {lead}: {f(bench.time_launches(lambda: ctx.decode(c, out), 7, 4)[0]):.3f}")
    ctx.set_option(capi.OPT_DECODE_READ_AHEAD_US, 0)
    print(f"bw {bw} exc {exc:2d} | " + " | ".join(row), flush=True)
    del c
