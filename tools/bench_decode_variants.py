#!/usr/bin/env python3
"""A/B micro-benchmark of the decode kernel variants (ALPGPU_DECODE_VARIANT) + HBM copy/fill calibration.
Usage on the GPU box: python tools/bench_decode_variants.py [n_vectors]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alp_amd import capi  # noqa: E402


def make_column(n, bw, exc_per_vec=0, seed=0, device=0):
    rng = np.random.default_rng(seed)
    vec = np.zeros(n, capi.VECTOR_DTYPE)
    vec["bw"] = bw
    vec["scheme"] = capi.SCHEME_ALP
    f = min(12, int((62 - bw) * 0.30103)) if bw <= 62 else 0
    vec["f"] = f
    vec["e"] = f + 2
    vec["base"] = rng.integers(0, 1 << max(1, min(bw, 62)), n)
    vec["packed_off"] = np.arange(n, dtype=np.uint64) * np.uint64(128 * bw)
    vec["exc_cnt"] = exc_per_vec
    rec = (10 * exc_per_vec + 7) // 8 * 8
    vec["exc_off"] = np.arange(n, dtype=np.uint64) * np.uint64(rec)
    rg = np.zeros((n + 99) // 100, capi.ROWGROUP_DTYPE)
    rg["scheme"] = capi.SCHEME_ALP
    rg["k"] = 1
    col = capi.DeviceColumn(n, device, packed_capacity=n * 128 * bw + 1024, exc_capacity=n * rec + 64)
    col.vectors.copy_(torch.from_numpy(vec.view(np.uint8).reshape(-1)))
    col.rowgroups[: rg.size * 32] = torch.from_numpy(rg.view(np.uint8).reshape(-1)).cuda()
    if bw:
        col.packed[: n * 128 * bw] = torch.randint(0, 256, (n * 128 * bw,), dtype=torch.uint8, device="cuda")
    if exc_per_vec:
        one = np.zeros(rec, np.uint8)
        one[: 8 * exc_per_vec] = rng.integers(0, 255, 8 * exc_per_vec)
        pos = np.sort(rng.choice(1024, exc_per_vec, replace=False)).astype(np.uint16)
        one[8 * exc_per_vec: 10 * exc_per_vec] = pos.view(np.uint8)
        col.exc[: n * rec] = torch.from_numpy(np.tile(one, n)).cuda()
    return col, rec


def timeit(fn, iters=5, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(iters)]
    return float(np.median(ts)), float(np.min(ts))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    out = torch.empty(n * 1024, dtype=torch.float64, device="cuda")
    src = torch.empty(n * 1024, dtype=torch.float64, device="cuda").normal_()
    ms, mn = timeit(lambda: out.copy_(src))
    print(f"calib copy  8B*{n*1024}: median {ms:.3f} ms  -> {2*n*8192/ms/1e9:.2f} TB/s read+write (best {2*n*8192/mn/1e9:.2f})")
    ms, mn = timeit(lambda: out.fill_(1.5))
    print(f"calib fill  : median {ms:.3f} ms -> {n*8192/ms/1e9:.2f} TB/s write-only (best {n*8192/mn/1e9:.2f})")
    del src
    ctxs = {}
    for variant in (1, 0, 4):
        os.environ["ALPGPU_DECODE_VARIANT"] = str(variant)
        ctxs[variant] = capi.Context(0)
    for bw, exc in ((8, 0), (16, 0), (16, 20), (28, 0), (28, 10), (40, 0)):
        col, rec = make_column(n, bw, exc, seed=bw)
        alg = n * (32 + 128 * bw + rec + 8192)
        line = f"bw={bw:2d} exc={exc:3d}  bytes/vec={alg//n}:"
        for variant, ctx in ctxs.items():
            ms, mn = timeit(lambda: ctx.decode(col, out))
            line += f"  v{variant}: {ms:.3f} ms {n*8192/ms/1e9:.2f} TB/s out, {alg/ms/1e9:.2f} TB/s traffic ({alg/ms/1e9/8.0*100:.0f}% of 8TB/s)"
        print(line, flush=True)
        del col
    print("device:", list(ctxs.values())[0].device_info())


if __name__ == "__main__":
    main()
