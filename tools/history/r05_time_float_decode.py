#!/usr/bin/env python3
"""r05_time_float_decode.py: alpgpu_decode_f32 on bench.py's two float columns and on a "clean" one (one decimal in +-100: few exceptions), at 1 / 2 / 4 vectors per
workgroup; ONE library per process (ALPGPU_LIB) — alternate processes for an A/B.  Fractions of 8 TB/s on algorithmic bytes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from alp_amd import capi  # noqa: E402

n = 1 << 20
VEC = 1024
dev = torch.device("cuda:0")
ctx = capi.Context(0)
outf = torch.empty(n * VEC, dtype=torch.float32, device=dev)
tag = os.path.basename(os.environ.get("ALPGPU_LIB", "libalpgpu.so"))
row = []
for kind in ("decimal_mixed", "rd", "clean"):
    g = torch.Generator(device=dev)
    g.manual_seed(43)
    if kind == "rd":
        xf = torch.rand(n * VEC, dtype=torch.float32, device=dev, generator=g)
    elif kind == "clean":
        xf = (torch.round((torch.rand(n * VEC, dtype=torch.float64, device=dev, generator=g) - 0.5) * 2e3) / 10.0).to(torch.float32)
    else:
        xd = (torch.rand(n * VEC, dtype=torch.float64, device=dev, generator=g) - 0.5) * 2e3
        sc = torch.where((torch.arange((n + 99) // 100, device=dev) % 2 == 0), 10.0, 100.0).to(torch.float64).repeat_interleave(100 * VEC)[: n * VEC]
        xf = (torch.round(xd * sc) / sc).to(torch.float32)
        m = torch.rand(n * VEC, device=dev, generator=g) < 0.01
        xf[m] = (xd[m] * 3.141592653589793).to(torch.float32)
        del xd, sc, m
    fcol = capi.DeviceColumn(n, 0, dtype="f32")
    ctx.encode(xf, fcol)
    pb, eb, ov = ctx.column_totals(fcol)
    alg = n * (4096 + 13) + pb + eb
    fr = []
    for vpw in (0, 1, 2, 4):
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
        med, _ = bench.time_launches(lambda: ctx.decode(fcol, outf), 9, 4)
        fr.append(alg / med / 1e6 / 8000)
    ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)
    fsums = torch.empty(n, dtype=torch.float64, device=dev)
    smed, _ = bench.time_launches(lambda: ctx.decode_sum(fcol, fsums), 9, 4)
    sfr = (alg - n * 4096 + n * 8) / smed / 1e6 / 8000
    del fsums
    rt = bool(torch.equal(outf.view(torch.int32), xf.view(torch.int32)))
    row.append(f"{kind} ({eb / n / 6:.0f} exc/vec, {pb * 8 / n / 1024:.1f} bits) auto {fr[0]:.3f} vpw1 {fr[1]:.3f} vpw2 {fr[2]:.3f} vpw4 {fr[3]:.3f} sum {sfr:.3f} rt {rt}")
    del xf, fcol
print(f"{tag} {bench.lib_sha16()}: " + " | ".join(row), flush=True)
