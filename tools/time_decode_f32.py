#!/usr/bin/env python3
"""time_decode_f32.py [n]: alpgpu_decode_f32 of bench.py's two float columns (run it under ALPGPU_DECODE_F32_PAD_LDS_KIB=... for the residency experiment)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from alp_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
dev = torch.device("cuda:0")
ctx = capi.Context(0)
ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, int(os.environ.get("SWEEP_VPW", "0")))
row = []
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
    col = capi.DeviceColumn(n, 0, dtype="f32")
    ctx.encode(xf, col)
    pb, eb, ov = ctx.column_totals(col)
    out = torch.empty(n * 1024, dtype=torch.float32, device=dev)
    best = 1e9
    for rnd in range(2):
        med, _ = bench.time_launches(lambda: ctx.decode(col, out), 7, 6)
        best = min(best, med)
    ok = bool(torch.equal(out.view(torch.int32), xf.view(torch.int32)))
    row.append(f"{kind}: {best:.3f} ms = {(n * (4096 + 13) + pb + eb) / best / 1e6 / 8000:.3f} of peak (round trip {ok})")
    del xf, col, out
print(f"float decode, vectors per workgroup {os.environ.get('SWEEP_VPW', 'auto')}, pad {os.environ.get('ALPGPU_DECODE_F32_PAD_LDS_KIB', '0'):>2s} KiB: " + " | ".join(row), flush=True)
