#!/usr/bin/env python3
"""prof_read_ahead.py [n]: the decode of long columns of narrow vectors under the library's own rule (for rocprofv3): 3 bits without exceptions (k_decode_column<1> with
k_read_ahead beside it), 4 bits with 20 exceptions per vector (k_decode_column<2> + k_read_ahead), and the same two with ALPGPU_OPT_DECODE_READ_AHEAD = 0.
(Under --pmc the profiler runs kernels one after the other: the read-ahead then waits its 50 ms for a decode that cannot start, leaves, and the decode runs cold —
the counters of that pass say nothing about the pair.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from alp_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
ctx = capi.Context(0)
out = torch.empty(n * 1024, dtype=torch.float64, device="cuda:0")
for bw, exc in ((3, 0), (4, 20)):
    c, _, ab = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=bw, exc_per_vec=exc)
    for opt in (-1, 0):
        ctx.set_option(capi.OPT_DECODE_READ_AHEAD, opt)
        med, _ = bench.time_launches(lambda: ctx.decode(c, out), 7, 3)
        print(f"bw {bw} exceptions {exc} read-ahead option {opt}: {med:.3f} ms = {ab / med / 1e6 / 8000:.3f} of peak ({ctx.decode_vectors_per_wg(c)} vector(s) per workgroup, "
              f"reads ahead: {ctx.decode_reads_ahead(c)})", flush=True)
    ctx.set_option(capi.OPT_DECODE_READ_AHEAD, -1)
    del c
