#!/usr/bin/env python3
"""time_decode_encoded.py: the store decode of what the ENCODER produces (bench.py's mixed / exc10 columns, the five real-data-shaped columns of tests/golden) in a grid of
launch shapes — vectors per workgroup x residency pad — next to the library's own choice after alpgpu_column_totals (decode of encoder-produced
columns >= 0.78).  Fractions of 8 TB/s over algorithmic bytes; profiles/r06_decode_policy.txt."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from alp_amd import capi  # noqa: E402

n = 1 << 20
ctx = capi.Context(0)
dev = torch.device("cuda:0")
out = torch.empty(n * 1024, dtype=torch.float64, device=dev)
print(f"lib {bench.lib_sha16()}")
z = np.load(os.path.join(ROOT, "tests", "golden", "rowgroup_samples.npz"))
cols = [("mixed", None), ("mixed_exc10", None)] + [(k, k) for k in sorted({k.split("__")[0] for k in z.files})]
pads = [int(p) for p in os.environ.get("PADS", "0,3,6,11,14").split(",")]
for label, gold in cols:
    if gold is None:
        x, nv = bench.synthetic_input(label, n, dev, seed=42), n
    else:
        nv = n // 100 * 100
        x = torch.from_numpy(z[gold + "__input_bits"][: 100 * 1024].view(np.int64)).to(dev).view(torch.float64).repeat(nv // 100)
    col = capi.DeviceColumn(nv, 0)
    ctx.encode(x, col)
    pb, eb, _ = ctx.column_totals(col)
    ab = bench.encode_alg_bytes(nv, pb, eb)
    f = lambda ms: ab / ms / 1e6 / 8000  # noqa: E731
    auto = f(bench.time_launches(lambda: ctx.decode(col, out), 7, 5)[0])
    row = [f"auto {auto:.3f} (vpw {ctx.decode_vectors_per_wg(col)}, ahead {int(ctx.decode_reads_ahead(col))}, runs {ctx.decode_runs(col)})"]
    for vpw in (1, 2):
        for pad in pads:
            ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
            ctx.set_option(capi.OPT_DECODE_RESIDENCY_PAD, pad)
            row.append(f"v{vpw}p{pad}:{f(bench.time_launches(lambda: ctx.decode(col, out), 7, 4)[0]):.3f}")
    ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)
    ctx.set_option(capi.OPT_DECODE_RESIDENCY_PAD, -1)
    ctx.set_option(capi.OPT_DECODE_READ_AHEAD, 1)
    row.append(f"ahead forced: {f(bench.time_launches(lambda: ctx.decode(col, out), 7, 4)[0]):.3f}")
    ctx.set_option(capi.OPT_DECODE_READ_AHEAD, -1)
    print(f"{label:28s} bits {(pb + eb) * 8 / (nv * 1024):5.2f} exc/vec {eb / 10 / nv:5.1f} | " + " ".join(row), flush=True)
    del x, col
