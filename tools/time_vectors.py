#!/usr/bin/env python3
"""time alpgpu_encode_vectors alone (states precomputed): time_vectors.py <mixed|rd|decimal|multik> [n_vectors]
multik: every vector has its own number of decimals (1..4), so rowgroups keep several (e,f) candidates and every vector runs
the second-level sampling (encoder.hpp:241-305)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alp_amd import capi
from bench import synthetic_input, time_launches
kind = sys.argv[1] if len(sys.argv) > 1 else "mixed"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
ctx = capi.Context(0)
if kind == "multik":
    g = torch.Generator(device="cuda:0"); g.manual_seed(5)
    dec = (torch.arange(n, device="cuda:0") * 7 % 4 + 1).to(torch.float64)
    sc = (10.0 ** dec).repeat_interleave(1024)
    x = torch.round((torch.rand(n * 1024, dtype=torch.float64, device="cuda:0", generator=g) - 0.5) * 2e3 * sc) / sc
    del sc
else:
    x = synthetic_input(kind, n, torch.device("cuda:0"), seed=42)
col = capi.DeviceColumn(n, 0, packed_capacity=n * 8448 + 1024 + n * 0, exc_capacity=n * 10240 + 64)
ctx.rowgroup_init(x, col)
med, _ = time_launches(lambda: ctx.encode_vectors(x, col), 5, 2)
rg = col.rowgroups.cpu().numpy().view(capi.ROWGROUP_DTYPE)[: col.n_rowgroups]
print("k histogram", {int(k): int((rg["k"] == k).sum()) for k in set(rg["k"].tolist())})
print(f"{kind}: encode_vectors median {med:.3f} ms for {n} vectors ({os.path.basename(os.environ.get('ALPGPU_LIB', 'libalpgpu.so'))})")
