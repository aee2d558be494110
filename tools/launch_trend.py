#!/usr/bin/env python3
"""per-launch decode time over a long run of back-to-back launches: is there a clock ramp after an idle gap?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from alp_amd import capi
from bench import build_decode_column, VEC
n = 1 << 20
ctx = capi.Context(0)
col, vec, ab = build_decode_column(n, 0, seed=42)
out = torch.empty(n * VEC, dtype=torch.float64, device="cuda")
for gap in (0.0, 0.5, 2.0):
    torch.cuda.synchronize(); time.sleep(gap)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(120)]
    for a, b in evs:
        a.record(); ctx.decode(col, out); b.record()
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in evs])
    print(f"idle {gap:.1f}s before: launches 0-2 {t[:3].round(3)}  3-9 mean {t[3:10].mean():.3f}  10-29 {t[10:30].mean():.3f}  30-59 {t[30:60].mean():.3f}  60-119 {t[60:].mean():.3f}  min {t.min():.3f}")
