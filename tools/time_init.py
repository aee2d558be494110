#!/usr/bin/env python3
"""time the rowgroup-init kernel alone: time_init.py <mixed|rd|decimal> [n_vectors] [f32]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alp_amd import capi
from bench import synthetic_input, time_launches
kind = sys.argv[1] if len(sys.argv) > 1 else "mixed"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
f32 = len(sys.argv) > 3
ctx = capi.Context(0)
x = synthetic_input(kind if kind != "decimal" else "mixed", n, torch.device("cuda:0"), seed=42)
if f32:
    x = x.to(torch.float32)
col = capi.DeviceColumn(n, 0, dtype="f32" if f32 else "f64")
med, _ = time_launches(lambda: ctx.rowgroup_init(x, col), 5, 2)
rg = col.rowgroups.cpu().numpy().view(capi.ROWGROUP_DTYPE)[: col.n_rowgroups]
print(f"{kind}{' f32' if f32 else ''}: n={n} rowgroup_init median {med:.3f} ms; ALP_RD rowgroups {(rg['scheme'] == 1).sum()} of {rg.size}; dict sizes {sorted(set(rg['rd_dict_size'].tolist()))}")
