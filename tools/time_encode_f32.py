#!/usr/bin/env python3
"""time_encode_f32.py [n]: alpgpu_encode_f32 on bench.py's float columns, rowgroup search beside / in front of the vector encode; bytes compared"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from alp_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
dev = torch.device("cuda:0")
ctx = capi.Context(0)
tag = os.path.basename(os.environ.get("ALPGPU_LIB", "libalpgpu.so"))
for kind in ("decimal_mixed", "rd"):
    g = torch.Generator(device=dev); g.manual_seed(43)
    if kind == "rd":
        xf = torch.rand(n * 1024, dtype=torch.float32, device=dev, generator=g)
    else:
        xd = (torch.rand(n * 1024, dtype=torch.float64, device=dev, generator=g) - 0.5) * 2e3
        sc = torch.where((torch.arange((n + 99) // 100, device=dev) % 2 == 0), 10.0, 100.0).to(torch.float64).repeat_interleave(100 * 1024)[: n * 1024]
        xf = (torch.round(xd * sc) / sc).to(torch.float32)
        m = torch.rand(n * 1024, device=dev, generator=g) < 0.01
        xf[m] = (xd[m] * 3.141592653589793).to(torch.float32)
        del xd, sc, m
    cols, ms = {}, {}
    for mode in (0, 1, 0, 1):
        ctx.set_option(capi.OPT_ENCODE_ASYNC_INIT, 2 if mode else 0)  # (float columns take the side search only with value 2)
        col = capi.DeviceColumn(n, 0, dtype="f32")
        ms[mode], _ = bench.time_launches(lambda: ctx.encode(xf, col), 5, 2)
        cols[mode] = col
    ctx.set_option(capi.OPT_ENCODE_ASYNC_INIT, 1)
    vmed, _ = bench.time_launches(lambda: ctx.encode_vectors(xf, cols[0]), 5, 2)
    pb, eb, ov = ctx.column_totals(cols[1])
    same = all(torch.equal(a, b) for a, b in ((cols[0].rowgroups, cols[1].rowgroups), (cols[0].vectors, cols[1].vectors), (cols[0].packed[:pb], cols[1].packed[:pb]), (cols[0].exc[:eb], cols[1].exc[:eb])))
    alg = n * (4096 + 13) + pb + eb
    print(f"{tag} f32 {kind} n={n}: search beside the encode {ms[1]:.3f} ms = {alg / ms[1] / 1e6 / 8000:.3f} of peak | in front {ms[0]:.3f} ms = {alg / ms[0] / 1e6 / 8000:.3f} | vectors alone {vmed:.3f} | same bytes: {same}", flush=True)
    del xf, cols
