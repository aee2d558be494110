#!/usr/bin/env python3
"""Experiment (needs a library built with -DALPGPU_FUSED_TIMING, tools/build_variant.sh timing -DALPGPU_FUSED_TIMING): where a
wavefront of k_encode_fused spends its life.  Every wavefront leaves 8 marks (10 ns ticks since it started):
0 input + rowgroup state arrived, 1 (e,f) chosen, 2 encoded, 3 staged + size posted, 4 packed, 5 exception image laid out,
6 ordered offset known, 7 stores issued.   usage: ALPGPU_LIB=.../libalpgpu_timing.so python tools/fused_phases.py [kind] [n]"""
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
for _ in range(3):
    ctx.encode_vectors(x, col)
ctx.synchronize()
buf = torch.zeros(n * 8, dtype=torch.int32, device="cuda")
capi.lib.alpgpu_debug_fused_phases.argtypes = [C.c_void_p]
assert capi.lib.alpgpu_debug_fused_phases(C.c_void_p(buf.data_ptr())) == 0
med, _ = time_launches(lambda: ctx.encode_vectors(x, col), 5, 1)
ctx.synchronize()
capi.lib.alpgpu_debug_fused_phases(C.c_void_p(0))
t = buf.view(n, 8).to(torch.float64) * 0.01  # microseconds
names = ["input + state arrive", "second-level (e,f)", "encode arithmetic", "stage + post size", "pack into registers",
         "exception image", "wait for ordered offset", "stores issued"]
print(f"{kind}: encode_vectors median {med:.3f} ms for {n} vectors WITH marks; phase lengths per wavefront, microseconds:")
prev = torch.zeros(n, dtype=torch.float64, device="cuda")
life = t[:, 7]
for k in range(8):
    d = t[:, k] - prev
    prev = t[:, k]
    q = lambda f: float(d.kthvalue(max(1, int(f * n))).values)
    print(f"  {names[k]:26s} mean {float(d.mean()):7.2f}  p10 {q(0.1):7.2f}  median {q(0.5):7.2f}  p90 {q(0.9):7.2f}   {100.0 * float(d.sum()) / float(life.sum()):5.1f} %")
print(f"  life of a wavefront        mean {float(life.mean()):7.2f}  median {float(life.median()):7.2f}")
w = t.view(-1, 8, 8)
print(f"  wavefront 0 (runs the look-back): wait {float((w[:, 0, 6] - w[:, 0, 5]).mean()):.2f}; others {float((w[:, 1:, 6] - w[:, 1:, 5]).mean()):.2f}")
print(f"  spread of 'size posted' inside a tile (max - min over its 8 wavefronts): mean {float((w[:, :, 3].max(dim=1).values - w[:, :, 3].min(dim=1).values).mean()):.2f}")
