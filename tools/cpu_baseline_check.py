#!/usr/bin/env python3
"""cpu_baseline_check.py: how the reference's CPU decode (oracle/_ref) scales with host threads on THIS box, on a DRAM-resident sample
(every thread its own 16 MiB slice) — the all-cores figure bench.py reports looked implausibly low on the first try (27 GB/s on 256 threads)."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
runner, kind, model = bench._cpu_runner()
print(kind, model, "cpus", len(os.sched_getaffinity(0)), flush=True)
per = 2048  # vectors per thread
rng = np.random.default_rng(1)
for stride_words in (1024, 1040):
    for nthreads in (1, 4, 16, 64, 128, 256):
        def prepare(t):
            bw = np.full(per, 1 + (t % 53), np.uint8)
            packed = rng.integers(0, 1 << 62, (per, stride_words), dtype=np.int64)
            return (packed, bw, np.full(per, 14, np.uint8), np.full(per, 12, np.uint8), np.zeros(per, np.int64), np.zeros(per, np.uint16), np.zeros((per, 8)), np.zeros((per, 8), np.uint16),
                    np.zeros(per * 1024))
        def work(t, m):
            return runner.time_falp_column(m[0], stride_words, m[1], m[2], m[3], m[4], m[5], m[6], m[7], 8, per, m[8], 20)
        wall, res = bench._run_threads(work, nthreads, prepare)
        print(f"stride {stride_words} words, {nthreads:3d} threads: {nthreads * per * 8192 * 20 / wall / 1e9:8.1f} GB/s decoded (wall {wall:.3f} s; slowest thread's own clock {max(res):.3f} s, fastest {min(res):.3f})", flush=True)
