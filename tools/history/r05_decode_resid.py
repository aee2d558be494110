#!/usr/bin/env python3
"""r05_decode_resid.py: ONE library (ALPGPU_LIB): per bit width, the store decode at one / two vectors per workgroup under each residency cap
(ALPGPU_OPT_DECODE_RESIDENCY_PAD: unused LDS per workgroup), without exceptions and with EXC (default 20) exceptions per vector (patch limit 64): the table the launch
rule of api.hip (decode_variant_for) is fitted to.  Fractions of 8 TB/s."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from alp_amd import capi  # noqa: E402

n = 1 << 20
widths = [int(w) for w in os.environ.get("WIDTHS", "1,2,3,4,6,8,10,12,14,16,18,20,24,28,32,36,40,44,48,53").split(",")]
excs = [int(e) for e in os.environ.get("EXCS", "0,20").split(",")]
pads = [int(p) for p in os.environ.get("PADS", "0,3,6,11,14,20").split(",")]
vpws = [int(v) for v in os.environ.get("VPWS", "1,2,4").split(",")]
ctx = capi.Context(0)
out = torch.empty(n * 1024, dtype=torch.float64, device="cuda:0")
tag = os.path.basename(os.environ.get("ALPGPU_LIB", "libalpgpu.so"))
print(f"{tag} {bench.lib_sha16()}: bw exc | vpw {vpws} each @ pads {pads} | best | auto (its vpw)", flush=True)
for exc in excs:
    for bw in widths:
        c, _, ab = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=bw, exc_per_vec=exc)
        rows = {}
        for vpw in vpws:
            ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
            rows[vpw] = []
            for pad in pads:
                ctx.set_option(capi.OPT_DECODE_RESIDENCY_PAD, pad)
                med, _ = bench.time_launches(lambda: ctx.decode(c, out), 5, 4)
                rows[vpw].append(ab / med / 1e6 / 8000)
        ctx.set_option(capi.OPT_DECODE_RESIDENCY_PAD, -1)
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)
        med, _ = bench.time_launches(lambda: ctx.decode(c, out), 5, 4)
        auto = ab / med / 1e6 / 8000
        best = max((f, v, p) for v in vpws for f, p in zip(rows[v], pads))
        print(f"{tag}: {bw:>2} {exc:>3} | " + " | ".join(" ".join(f"{f:.3f}" for f in rows[v]) for v in vpws) + f" | {best[0]:.3f} vpw{best[1]} pad{best[2]} | {auto:.3f} ({ctx.decode_vectors_per_wg(c)})", flush=True)
        del c
