#!/usr/bin/env python3
"""Does it matter WHERE a vector's exception record lives?  Same kernel, two layouts of the same column:
  separate : packed stream and exception stream are two buffers (the format of include/alpgpu.h)
  inline   : each vector's exception record sits right behind its packed words in ONE buffer (d_exc == d_packed)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from alp_amd import capi
from bench_decode_variants import timeit

def make(n, bw, exc, inline):
    rng = np.random.default_rng(1)
    rec = (10 * exc + 15) // 16 * 16
    vec = np.zeros(n, capi.VECTOR_DTYPE)
    vec["bw"], vec["scheme"], vec["e"], vec["f"], vec["exc_cnt"] = bw, capi.SCHEME_ALP, 14, 12, exc
    vec["base"] = rng.integers(0, 1 << bw, n)
    stride = 128 * bw + (rec if inline else 0)
    vec["packed_off"] = np.arange(n, dtype=np.uint64) * np.uint64(stride)
    vec["exc_off"] = vec["packed_off"] + np.uint64(128 * bw) if inline else np.arange(n, dtype=np.uint64) * np.uint64(rec)
    col = capi.DeviceColumn(n, 0, packed_capacity=n * stride + 1024, exc_capacity=(64 if inline else n * rec + 64))
    col.vectors.copy_(torch.from_numpy(vec.view(np.uint8).reshape(-1)))
    rg = np.zeros((n + 99) // 100, capi.ROWGROUP_DTYPE); rg["scheme"] = capi.SCHEME_ALP; rg["k"] = 1
    col.rowgroups[: rg.size * 32] = torch.from_numpy(rg.view(np.uint8).reshape(-1)).cuda()
    col.packed[: n * stride] = torch.randint(0, 256, (n * stride,), dtype=torch.uint8, device="cuda")
    one = np.zeros(rec, np.uint8)
    one[: 8 * exc] = rng.integers(0, 255, 8 * exc)
    one[8 * exc: 10 * exc] = np.sort(rng.choice(1024, exc, replace=False)).astype(np.uint16).view(np.uint8)
    if inline:
        pv = col.packed[: n * stride].view(n, stride)
        pv[:, 128 * bw:] = torch.from_numpy(one).cuda()
        col.c.d_exc = col.c.d_packed
        col.c.exc_capacity = col.c.packed_capacity
    else:
        col.exc[: n * rec] = torch.from_numpy(np.tile(one, n)).cuda()
    return col

n = 1 << 20
out = torch.empty(n * 1024, dtype=torch.float64, device="cuda")
for vpw in (1, 2):
    ctx = capi.Context(0); ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
    for bw, exc in ((16, 5), (16, 20), (28, 10)):
        line = f"vectors/wg={vpw} bw={bw} exc={exc}:"
        for inline in (False, True):
            col = make(n, bw, exc, inline)
            ms, mn = timeit(lambda: ctx.decode(col, out), 7, 2)
            alg = n * (128 * bw + 10 * exc + 13 + 8192)
            line += f"  {'inline  ' if inline else 'separate'} {ms:.3f} ms {alg/ms/1e9/8*100:.0f}%"
            del col
        print(line, flush=True)
