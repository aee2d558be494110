#!/usr/bin/env python3
"""r05_hit_rate.py: is the read-ahead's residual gap to the warm level (0.90 at one vector per workgroup, six workgroups per CU) a matter of timing / eviction or of the two kernels
running side by side?  256 Ki-vector columns whose whole input fits the Infinity Cache, COLD (2 GiB read between launches): option off; read-ahead with the lead by width; read-ahead
with a lead of 10 ms (everything is read at once, nothing can arrive late or be evicted); and warm (back to back, option off).  One vector per workgroup, pads 0 / 14."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from alp_amd import capi  # noqa: E402

n = 1 << 18
ctx = capi.Context(0)
out = torch.empty(n * 1024, dtype=torch.float64, device="cuda:0")
flush = torch.ones(1 << 29, dtype=torch.float32, device="cuda:0")


def cold(fn):
    ts = []
    for _ in range(9):
        flush.sum()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


print(f"lib {bench.lib_sha16()}: bw exc pad | cold off | cold, lead by width | cold, lead 10 ms | warm off", flush=True)
for bw, exc in ((2, 0), (4, 0), (4, 20)):
    c, _, ab = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=bw, exc_per_vec=exc)
    fr = lambda ms: ab / ms / 1e6 / 8000  # noqa: E731
    ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 2 if exc else 1)
    for pad in (0, 14):
        ctx.set_option(capi.OPT_DECODE_RESIDENCY_PAD, pad)
        ctx.set_option(capi.OPT_DECODE_READ_AHEAD, 0)
        off = cold(lambda: ctx.decode(c, out))
        warm, _ = bench.time_launches(lambda: ctx.decode(c, out), 9, 4)
        ctx.set_option(capi.OPT_DECODE_READ_AHEAD, 1)
        ctx.set_option(capi.OPT_DECODE_READ_AHEAD_US, 0)
        by_width = cold(lambda: ctx.decode(c, out))
        ctx.set_option(capi.OPT_DECODE_READ_AHEAD_US, 10000)
        at_once = cold(lambda: ctx.decode(c, out))
        ctx.set_option(capi.OPT_DECODE_READ_AHEAD_US, 0)
        print(f"{bw:>2} {exc:>2} pad {pad:>2} | {fr(off):.3f} | {fr(by_width):.3f} | {fr(at_once):.3f} | {fr(warm):.3f}", flush=True)
    del c
