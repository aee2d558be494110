#!/usr/bin/env python3
"""time_encode.py [n_vectors] [kinds...]: alpgpu_encode_f64 ordered (look-back) against ALPGPU_OPT_ENCODE_UNORDERED (one atomic add per tile), arms
alternating; the vector encode alone in both forms; the traffic probe alone and with the persistent search beside it; a read-only stream.
profiles/r0N_encode_levers.txt."""
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
dev = torch.device("cuda:0")
out = torch.empty(n * 1024, dtype=torch.float64, device=dev)
print(f"lib {bench.lib_sha16()}  n={n}  env: " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("ALPGPU_")))
for kind in kinds:
    x = bench.synthetic_input(kind, n, dev, seed=42)
    col = capi.DeviceColumn(n, 0)
    ms = {0: [], 1: []}
    vs = {0: [], 1: []}
    for rep in range(3):
        for u in (0, 1):
            ctx.set_option(capi.OPT_ENCODE_UNORDERED, u)
            ms[u].append(bench.time_launches(lambda: ctx.encode(x, col), 7, 3)[0])
    ctx.set_option(capi.OPT_ENCODE_UNORDERED, 0)
    ctx.encode(x, col)
    pb, eb, ov = ctx.column_totals(col)
    alg = bench.encode_alg_bytes(n, pb, eb)
    for u in (0, 1):
        ctx.set_option(capi.OPT_ENCODE_UNORDERED, u)
        vs[u].append(bench.time_launches(lambda: ctx.encode_vectors(x, col), 5, 2)[0])
    ctx.set_option(capi.OPT_ENCODE_UNORDERED, 1)
    ctx.encode(x, col)
    ctx.decode(col, out)
    torch.cuda.synchronize()
    rt = torch.equal(out.view(torch.int64), x.view(torch.int64))
    ctx.set_option(capi.OPT_ENCODE_UNORDERED, 0)
    imed = bench.time_launches(lambda: ctx.rowgroup_init(x, col), 5, 2)[0]
    wb = (pb + eb + 13 * n) // n // 16 * 16
    p = bench.time_launches(lambda: ctx.traffic_probe(x, out, n, wb), 7, 3)[0]
    ps = bench.time_launches(lambda: ctx.traffic_probe_with_search(x, out, n, wb, col), 7, 3)[0]
    ro = bench.time_launches(lambda: ctx.traffic_probe(x, out, n, 0), 7, 3)[0]  # (write_bytes 0: a read-only stream, 8 bytes per vector stored)
    f = lambda t: alg / t / 1e6 / 8000  # noqa: E731
    print(f"{kind}: ordered {' '.join(f'{t:.3f}' for t in ms[0])} ms = {f(min(ms[0])):.3f}-{f(max(ms[0])):.3f} | unordered {' '.join(f'{t:.3f}' for t in ms[1])} ms = {f(min(ms[1])):.3f}-{f(max(ms[1])):.3f} "
          f"(round trip {rt}) | vectors alone: ordered {vs[0][0]:.3f} unordered {vs[1][0]:.3f} | search alone {imed:.3f} | probe ({wb} B written per vector) {p:.3f} ms = {f(p):.3f}, with the search beside {ps:.3f} ms = {f(ps):.3f} "
          f"| read-only 8 KiB per vector {ro:.3f} ms = {n * 8192 / ro / 1e6 / 8000:.3f} of peak", flush=True)
    del x, col
