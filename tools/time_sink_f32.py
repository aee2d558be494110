#!/usr/bin/env python3
"""time_sink_f32.py [n]: alpgpu_decode_sum_f32 on bench.py's float columns (one wavefront per vector), with the columns' width / exception profile;
ALPGPU_LIB selects an A/B build (e.g. -DALPGPU_SINK_STAGE_F32=0: no LDS stage; -DALPGPU_SINK_STAGE_F32_MAX_EXC=n)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
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
        sp = torch.rand(n * 1024, device=dev, generator=g) < 0.001
        specials = torch.tensor([float("nan"), float("inf"), float("-inf"), -0.0], dtype=torch.float32, device=dev)
        xf[sp] = specials[torch.randint(0, 4, (int(sp.sum()),), device=dev, generator=g)]
        del xd, sc, m, sp
    col = ctx.encode(xf)
    pb, eb, ov = ctx.column_totals(col)
    vec = col.vectors[: min(n, 20000) * 32].cpu().numpy().view(capi.VECTOR_DTYPE)
    sums = torch.empty(n, dtype=torch.float64, device=dev)
    med, _ = bench.time_launches(lambda: ctx.decode_sum(col, sums), 9, 6)
    alg = n * 13 + pb + eb + 8 * n
    print(f"{tag} f32 {kind}: decode_sum {med:.3f} ms = {alg / med / 1e6 / 8000:.3f} of peak | bw p10/50/90 {np.percentile(vec['bw'], [10, 50, 90])} exceptions p10/50/90 {np.percentile(vec['exc_cnt'], [10, 50, 90])} "
          f"scheme ALP share {float((vec['scheme'] == 2).mean()):.2f}", flush=True)
    del xf, col
