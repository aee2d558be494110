#!/usr/bin/env python3
"""run the float encode path a few times (for rocprofv3): prof_encode_f32.py <decimal2|decimal1|rd> [n_vectors]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alp_amd import capi
kind = sys.argv[1] if len(sys.argv) > 1 else "rd"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
ctx = capi.Context(0)
torch.manual_seed(3)
if kind == "decimal2":
    x = (torch.round(torch.rand(n * 1024, device="cuda", dtype=torch.float64) * 1e5) / 100).to(torch.float32)
elif kind == "decimal1":
    x = torch.where(torch.rand(n * 1024, device="cuda") < 0.01, torch.rand(n * 1024, device="cuda") * 3.14159,
                    (torch.round(torch.rand(n * 1024, device="cuda", dtype=torch.float64) * 1e4) / 10).to(torch.float32))
else:
    x = torch.rand(n * 1024, device="cuda", dtype=torch.float32)
col = capi.DeviceColumn(n, 0, dtype="f32")
for _ in range(5):
    ctx.encode(x, col)
ctx.synchronize()
import numpy as np
rg = col.rowgroups.cpu().numpy().view(capi.ROWGROUP_DTYPE)[: col.n_rowgroups]
print(kind, "schemes", {int(k): int((rg["scheme"] == k).sum()) for k in set(rg["scheme"].tolist())}, "k", {int(k): int((rg["k"] == k).sum()) for k in set(rg["k"].tolist())}, ctx.column_totals(col))
