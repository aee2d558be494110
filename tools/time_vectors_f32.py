#!/usr/bin/env python3
"""time alpgpu_encode_vectors_f32 alone (states precomputed) on bench.py's float columns: time_vectors_f32.py <decimal_mixed|rd> [n_vectors]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alp_amd import capi
from bench import time_launches
kind = sys.argv[1] if len(sys.argv) > 1 else "decimal_mixed"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
dev = torch.device("cuda:0")
ctx = capi.Context(0)
g = torch.Generator(device=dev); g.manual_seed(43)
xd = (torch.rand(n * 1024, dtype=torch.float64, device=dev, generator=g) - 0.5) * 2e3
if kind == "rd":
    xf = (xd * 3.141592653589793).to(torch.float32)
else:
    sc = torch.where((torch.arange((n + 99) // 100, device=dev) % 2 == 0), 10.0, 100.0).to(torch.float64).repeat_interleave(100 * 1024)[: n * 1024]
    xf = (torch.round(xd * sc) / sc).to(torch.float32)
    m = torch.rand(n * 1024, device=dev, generator=g) < 0.01
    xf[m] = (xd[m] * 3.141592653589793).to(torch.float32)
    del sc, m
del xd
col = capi.DeviceColumn(n, 0, dtype="f32")
ctx.rowgroup_init(xf, col)
med, _ = time_launches(lambda: ctx.encode_vectors(xf, col), 5, 2)
imed, _ = time_launches(lambda: ctx.rowgroup_init(xf, col), 5, 1)
pb, eb, ov = ctx.column_totals(col)
out = torch.empty(n * 1024, dtype=torch.float32, device=dev)
ctx.decode(col, out); ctx.synchronize()
print(f"{kind}: encode_vectors_f32 median {med:.3f} ms, rowgroup_init {imed:.3f} ms for {n} vectors; {(pb + eb) * 8 / (n * 1024):.2f} bits/value; "
      f"roundtrip {bool(torch.equal(out.view(torch.int32), xf.view(torch.int32)))} ({os.path.basename(os.environ.get('ALPGPU_LIB', 'libalpgpu.so'))})")
