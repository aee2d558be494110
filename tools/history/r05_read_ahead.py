#!/usr/bin/env python3
"""r05_read_ahead.py: the store decode with ALPGPU_OPT_DECODE_READ_AHEAD (read_ahead_kernels.hip) against the plain launch, 1 Mi-vector columns of one bit width
(and the benchmark column, bw = "mix"), without and with EXC exceptions per vector: per vectors-per-workgroup x residency pad; then the window.
Fractions of 8 TB/s on algorithmic bytes.  Env: WIDTHS, EXCS, PADS, VPWS, WINDOWS (MiB; first = the one used in the shape table)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from alp_amd import capi  # noqa: E402

n = 1 << 20
widths = [w for w in os.environ.get("WIDTHS", "2,4,8,12,16,20,28,36,44,53,mix").split(",")]
excs = [int(e) for e in os.environ.get("EXCS", "0,20").split(",")]
pads = [int(p) for p in os.environ.get("PADS", "0,6,14").split(",")]
vpws = [int(v) for v in os.environ.get("VPWS", "1,2").split(",")]
windows = [int(v) for v in os.environ.get("WINDOWS", "40,20,80").split(",")]
ctx = capi.Context(0)
out = torch.empty(n * 1024, dtype=torch.float64, device="cuda:0")
print(f"lib {bench.lib_sha16()} grid {os.environ.get('ALPGPU_READ_AHEAD_GRID', 'default')}: bw exc | plain auto (vpw) | read-ahead {windows[0]} us: vpw {vpws} each @ pads {pads} | best | "
      f"best shape at windows {windows[1:]} | read-ahead with the rule's own shape", flush=True)


def frac(ab):
    med, _ = bench.time_launches(lambda: ctx.decode(c, out), 7, 4)
    return ab / med / 1e6 / 8000


for exc in excs:
    for w in widths:
        if w == "mix" and exc:
            continue
        c, _, ab = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=None if w == "mix" else int(w), exc_per_vec=exc)
        ctx.set_option(capi.OPT_DECODE_READ_AHEAD, 0)
        ctx.set_option(capi.OPT_DECODE_RESIDENCY_PAD, -1)
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)
        plain = frac(ab)
        auto_vpw = ctx.decode_vectors_per_wg(c)
        ctx.set_option(capi.OPT_DECODE_READ_AHEAD, 1)
        ctx.set_option(capi.OPT_DECODE_READ_AHEAD_US, windows[0])
        own = frac(ab)
        rows = {}
        for vpw in vpws:
            ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
            rows[vpw] = []
            for pad in pads:
                ctx.set_option(capi.OPT_DECODE_RESIDENCY_PAD, pad)
                rows[vpw].append(frac(ab))
        best = max((f, v, p) for v in vpws for f, p in zip(rows[v], pads))
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, best[1])
        ctx.set_option(capi.OPT_DECODE_RESIDENCY_PAD, best[2])
        wins = []
        for mib in windows[1:]:
            ctx.set_option(capi.OPT_DECODE_READ_AHEAD_US, mib)
            wins.append(frac(ab))
        print(f"{w:>3} {exc:>3} | {plain:.3f} ({auto_vpw}) | " + " | ".join(" ".join(f"{f:.3f}" for f in rows[v]) for v in vpws)
              + f" | {best[0]:.3f} vpw{best[1]} pad{best[2]} | " + " ".join(f"{f:.3f}" for f in wins) + f" | {own:.3f}", flush=True)
        del c
