#!/usr/bin/env python3
"""r05_decode_ab.py: ONE library (ALPGPU_LIB) over a few widths: the exception-free column at one / two vectors per workgroup, and the column with 20
exceptions per vector through the mask (patch limit 0) and through the patch arm (limit 64), at fixed launch shapes.  Run once per A/B library
in the same gpurun call (boxes differ).  An old library (round 4) ignores the patch option: its rows are the mask route's."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from alp_amd import capi  # noqa: E402

n = 1 << 20
widths = [int(w) for w in os.environ.get("WIDTHS", "1,3,6,12,20,32,44").split(",")]
ctx = capi.Context(0)
out = torch.empty(n * 1024, dtype=torch.float64, device="cuda:0")
tag = os.path.basename(os.environ.get("ALPGPU_LIB", "libalpgpu.so"))
old = "r04" in tag


def opt(o, v):
    try:
        ctx.set_option(o, v)
    except capi.AlpGpuError:
        pass


def t(col, ab, patch, vpw, pairing=0, plain=0):
    opt(capi.OPT_DECODE_PATCH_AFTER, patch)
    opt(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
    opt(capi.OPT_DECODE_PAIRING, pairing)
    opt(capi.OPT_DECODE_PLAIN_STORES, plain)
    med, _ = bench.time_launches(lambda: ctx.decode(col, out), 7, 5)
    opt(capi.OPT_DECODE_PLAIN_STORES, 0)
    opt(capi.OPT_DECODE_PAIRING, 0)
    return ab / med / 1e6 / 8000


print(f"{tag}: bw | exc0: vpw1 vpw2 auto | exc20 mask: vpw1 vpw2 pairs | exc20 patch: vpw1 vpw2 vpw2+plain-stores auto", flush=True)
for bw in widths:
    c0, _, ab0 = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=bw, exc_per_vec=0)
    r0 = [t(c0, ab0, 64, 1), t(c0, ab0, 64, 2), t(c0, ab0, 64, 0)]
    del c0
    c, _, ab = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=bw, exc_per_vec=20)
    rm = [t(c, ab, 0, 1), t(c, ab, 0, 2), t(c, ab, 0, 0, pairing=1)]
    rp = [] if old else [t(c, ab, 64, 1), t(c, ab, 64, 2), t(c, ab, 64, 2, plain=1), t(c, ab, 64, 0)]
    del c
    print(f"{tag}: {bw:>2} | " + " ".join(f"{f:.3f}" for f in r0) + " | " + " ".join(f"{f:.3f}" for f in rm) + " | " + " ".join(f"{f:.3f}" for f in rp), flush=True)
