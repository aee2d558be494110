#!/usr/bin/env python3
"""time_f32_sum.py [n]: bench.py's two float columns (decimal_mixed, rd) — SUM by the one-wavefront kernel, by the staged one, and the store decode; fractions of
8 TB/s over algorithmic bytes (bench.py: float_path).  For A/B libraries (ALPGPU_LIB=...)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from alp_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
VEC = 1024
dev = "cuda:0"
ctx = capi.Context(0)
print(f"lib {bench.lib_sha16()}")
outf = torch.empty(n * VEC, dtype=torch.float32, device=dev)
for kind in ("decimal_mixed", "rd"):
    g = torch.Generator(device=dev)
    g.manual_seed(43)
    if kind == "rd":
        xf = torch.rand(n * VEC, dtype=torch.float32, device=dev, generator=g)
    else:
        xd = (torch.rand(n * VEC, dtype=torch.float64, device=dev, generator=g) - 0.5) * 2e3
        sc = torch.where((torch.arange((n + 99) // 100, device=dev) % 2 == 0), 10.0, 100.0).to(torch.float64).repeat_interleave(100 * VEC)[: n * VEC]
        xf = (torch.round(xd * sc) / sc).to(torch.float32)
        m = torch.rand(n * VEC, device=dev, generator=g) < 0.01
        xf[m] = (xd[m] * 3.141592653589793).to(torch.float32)
        sp = torch.rand(n * VEC, device=dev, generator=g) < 0.001
        specials = torch.tensor([float("nan"), float("inf"), float("-inf"), -0.0], dtype=torch.float32, device=dev)
        xf[sp] = specials[torch.randint(0, 4, (int(sp.sum()),), device=dev, generator=g)]
        del xd, sc, m, sp
    fcol = capi.DeviceColumn(n, 0, dtype="f32")
    ctx.encode(xf, fcol)
    pb, eb, ov = ctx.column_totals(fcol)
    f_alg = n * (4096 + 13) + pb + eb
    fr = lambda b, ms: b / ms / 1e6 / 8000  # noqa: E731
    dmed, _ = bench.time_launches(lambda: ctx.decode(fcol, outf), 7, 10)
    rt = bool(torch.equal(outf.view(torch.int32), xf.view(torch.int32)))
    fsums = torch.empty(n, dtype=torch.float64, device=dev)
    smed, _ = bench.time_launches(lambda: ctx.decode_sum(fcol, fsums), 7, 10)
    ref = fsums.clone()
    ctx.set_option(capi.OPT_CONSUMER_PIPELINED, 3)
    s4, _ = bench.time_launches(lambda: ctx.decode_sum(fcol, fsums), 7, 5)
    same = bool(torch.equal(torch.nan_to_num(ref).view(torch.int64), torch.nan_to_num(fsums).view(torch.int64)))
    ctx.set_option(capi.OPT_CONSUMER_PIPELINED, 0)
    print(f"{kind:14s} dec {fr(f_alg, dmed):.3f} (rt {rt})  sum {smed:.3f} ms = {fr(f_alg - n * 4096 + n * 8, smed):.3f}  staged sum {s4:.3f} ms = {fr(f_alg - n * 4096 + n * 8, s4):.3f} (same bits {same})", flush=True)
    del xf, fcol, fsums
