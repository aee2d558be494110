#!/usr/bin/env python3
"""r05_beside.py: what a second, nearly idle kernel on another stream costs the store decode.  ONE process per configuration (the read-ahead's mode / grid /
stream priority come from the environment): the benchmark column and a 4-bit column, plain launch, then with the read-ahead option on (whatever
ALPGPU_READ_AHEAD_MODE makes of it), then plain with torch.cuda._sleep on a side stream (one thread spinning)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from alp_amd import capi  # noqa: E402

n = 1 << 20
ctx = capi.Context(0)
out = torch.empty(n * 1024, dtype=torch.float64, device="cuda:0")
side = torch.cuda.Stream(device="cuda:0")
tag = f"mode {os.environ.get('ALPGPU_READ_AHEAD_MODE', '0')} grid {os.environ.get('ALPGPU_READ_AHEAD_GRID', '128')} prio {os.environ.get('ALPGPU_INIT_STREAM_PRIO', 'high')}"
for w in (None, 4, 36):
    c, _, ab = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=w, exc_per_vec=0)

    def frac(fn):
        med, _ = bench.time_launches(fn, 9, 4)
        return ab / med / 1e6 / 8000

    ctx.set_option(capi.OPT_DECODE_READ_AHEAD, 0)
    plain = frac(lambda: ctx.decode(c, out))
    ctx.set_option(capi.OPT_DECODE_READ_AHEAD, 1)
    beside = frac(lambda: ctx.decode(c, out))
    ctx.set_option(capi.OPT_DECODE_READ_AHEAD, 0)

    def with_sleep():
        with torch.cuda.stream(side):
            torch.cuda._sleep(3_000_000)  # ~1.3 ms of one spinning thread
        ctx.decode(c, out)

    sl = frac(with_sleep)
    torch.cuda.synchronize()
    print(f"{tag}: bw {str(w):>4} | plain {plain:.3f} | option on {beside:.3f} | torch._sleep beside {sl:.3f}", flush=True)
    del c
