#!/usr/bin/env python3
"""r05_sinkf_counters.py <clean|mixed|rd> [n]: five launches of the one-wavefront float SUM sink (k_sink_direct_f32) on one column, for tools/pmc_busy.sh under
ALPGPU_LIB = a -DALPGPU_SINKF_STOP_AT=n build: the counters of successive builds difference into instructions per stage (profiles/r05_float_sink.txt).
clean: one decimal, no exceptions; mixed / rd: bench.py's two float columns."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from alp_amd import capi  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "clean"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
dev = torch.device("cuda:0")
ctx = capi.Context(0)
g = torch.Generator(device=dev)
g.manual_seed(43)
if kind == "rd":
    xf = torch.rand(n * 1024, dtype=torch.float32, device=dev, generator=g)
else:
    xd = (torch.rand(n * 1024, dtype=torch.float64, device=dev, generator=g) - 0.5) * 2e3
    if kind == "clean":
        xf = (torch.round(xd * 10.0) / 10.0).to(torch.float32)
    else:
        sc = torch.where((torch.arange((n + 99) // 100, device=dev) % 2 == 0), 10.0, 100.0).to(torch.float64).repeat_interleave(100 * 1024)[: n * 1024]
        xf = (torch.round(xd * sc) / sc).to(torch.float32)
        m = torch.rand(n * 1024, device=dev, generator=g) < 0.01
        xf[m] = (xd[m] * 3.141592653589793).to(torch.float32)
col = ctx.encode(xf)
ctx.synchronize()
import numpy as np  # noqa: E402
vec = col.vectors.cpu().numpy().view(capi.VECTOR_DTYPE)[:n]
print(f"{kind}: ALP share {(vec['scheme'] == 2).mean():.3f}  bw p10/50/90 {np.percentile(vec['bw'], [10, 50, 90])}  exceptions p10/50/90 {np.percentile(vec['exc_cnt'], [10, 50, 90])}  "
      f"vectors with > 128 exceptions {(vec['exc_cnt'] > 128).mean():.3f}, with 0: {(vec['exc_cnt'] == 0).mean():.3f}", flush=True)
sums = torch.empty(n, dtype=torch.float64, device=dev)
ctx.set_option(capi.OPT_CONSUMER_PIPELINED, 2)
for _ in range(5):
    ctx.decode_sum(col, sums)
torch.cuda.synchronize()
