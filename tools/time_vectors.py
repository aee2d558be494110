#!/usr/bin/env python3
"""time alpgpu_encode_vectors alone (states precomputed): time_vectors.py <mixed|rd|decimal> [n_vectors]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alp_amd import capi
from bench import synthetic_input, time_launches
kind = sys.argv[1] if len(sys.argv) > 1 else "mixed"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
ctx = capi.Context(0)
x = synthetic_input(kind, n, torch.device("cuda:0"), seed=42)
col = capi.DeviceColumn(n, 0, packed_capacity=n * 8448 + 1024 + n * 0, exc_capacity=n * 10240 + 64)
ctx.rowgroup_init(x, col)
med, _ = time_launches(lambda: ctx.encode_vectors(x, col), 5, 2)
print(f"{kind}: encode_vectors median {med:.3f} ms for {n} vectors ({os.path.basename(os.environ.get('ALPGPU_LIB', 'libalpgpu.so'))})")
