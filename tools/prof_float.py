#!/usr/bin/env python3
"""run the float encode + decode a few times (for rocprofv3): prof_float.py [n_vectors]  (bench.py's decimal_mixed float column)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alp_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
dev = torch.device("cuda:0")
ctx = capi.Context(0)
g = torch.Generator(device=dev); g.manual_seed(43)
xd = (torch.rand(n * 1024, dtype=torch.float64, device=dev, generator=g) - 0.5) * 2e3
sc = torch.where((torch.arange((n + 99) // 100, device=dev) % 2 == 0), 10.0, 100.0).to(torch.float64).repeat_interleave(100 * 1024)[: n * 1024]
xf = (torch.round(xd * sc) / sc).to(torch.float32)
m = torch.rand(n * 1024, device=dev, generator=g) < 0.01
xf[m] = (xd[m] * 3.141592653589793).to(torch.float32)
del xd, sc, m
col = capi.DeviceColumn(n, 0, dtype="f32")
out = torch.empty(n * 1024, dtype=torch.float32, device=dev)
for _ in range(6):
    ctx.encode(xf, col)
ctx.synchronize()
if os.environ.get("ALPGPU_PROF_ENCODE_ONLY"):  # measurement builds that write garbage columns: nothing may read what they wrote
    sys.exit(0)
print(ctx.column_totals(col))
for _ in range(12):
    ctx.decode(col, out)
ctx.synchronize()
print("roundtrip", bool(torch.equal(out.view(torch.int32), xf.view(torch.int32))))
