#!/usr/bin/env python3
"""Experiment (needs alp_amd/libalpgpu_timing.so built with -DALPGPU_FUSED_TIMING): average cycles a wavefront of
k_encode_fused spends in each phase.  usage: ALPGPU_LIB=.../libalpgpu_timing.so python tools/fused_phases.py [kind] [n]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alp_amd import capi
from bench import synthetic_input, time_launches
kind = sys.argv[1] if len(sys.argv) > 1 else "mixed"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
ctx = capi.Context(0)
x = synthetic_input(kind, n, torch.device("cuda:0"), seed=42)
col = capi.DeviceColumn(n, 0)
ctx.rowgroup_init(x, col)
ctx.encode_vectors(x, col); ctx.synchronize()
buf = torch.zeros(n * 4, dtype=torch.int32, device="cuda")
capi.lib.alpgpu_debug_fused_phases.argtypes = [C.c_void_p]
capi.lib.alpgpu_debug_fused_phases(C.c_void_p(buf.data_ptr()))
lb = torch.zeros((n + 3) // 4 * 4, dtype=torch.int32, device="cuda")
if hasattr(capi.lib, "alpgpu_debug_lookback_phases"):
    capi.lib.alpgpu_debug_lookback_phases.argtypes = [C.c_void_p]
    capi.lib.alpgpu_debug_lookback_phases(C.c_void_p(lb.data_ptr()))
med, _ = time_launches(lambda: ctx.encode_vectors(x, col), 5, 1)
ctx.synchronize()
t = buf.view(n, 4).to(torch.float64)
names = ["input load", "encode + stage", "ordered offset (look-back / wait)", "records + pack + stores"]
tot = float(t.sum())
print(f"{kind}: encode_vectors median {med:.3f} ms for {n} vectors; per-wavefront phase lengths (shader clock ticks):")
for k in range(4):
    col_k = t[:, k]
    print(f"  {names[k]:36s} mean {float(col_k.mean()):9.1f}  median {float(col_k.median()):9.1f}  p95 {float(col_k.kthvalue(int(0.95 * n)).values):9.1f}   {100.0 * float(col_k.sum()) / tot:5.1f} %")
w0 = t.view(-1, 4, 4)[:, 0, 2]
print(f"  look-back of wavefront 0 alone: mean {float(w0.mean()):.1f}")
L = lb.view(-1, 4).to(torch.float64)
for k, nm in enumerate(["level-1 ticks", "level-1 rounds", "level-2 ticks", "level-2 rounds"]):
    c = L[:, k]
    print(f"  look-back {nm:16s} mean {float(c.mean()):9.1f}  median {float(c.median()):9.1f}  p95 {float(c.kthvalue(int(0.95 * c.numel())).values):9.1f}")
