#!/usr/bin/env python3
"""run the encode path a few times (for rocprofv3): prof_encode.py <mixed|rd|decimal> [n_vectors]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alp_amd import capi
sys.argv = [sys.argv[0]] + sys.argv[1:]
from bench import synthetic_input, time_launches
kind = sys.argv[1] if len(sys.argv) > 1 else "mixed"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
ctx = capi.Context(0)
dev = torch.device("cuda:0")
if kind == "decimal":
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = torch.round((torch.rand(n * 1024, dtype=torch.float64, device=dev, generator=g) - 0.5) * 2e7) / 100.0
else:
    x = synthetic_input(kind, n, dev, seed=42)
col = capi.DeviceColumn(n, 0)
med, mean = time_launches(lambda: ctx.encode(x, col), 5, 2)
if os.environ.get("ALPGPU_PROF_ENCODE_ONLY"):  # measurement builds that write garbage columns (-DALPGPU_LEAN_STOP_AT): nothing may read what they wrote
    print(f"{kind}: n={n} encode median {med:.3f} ms (encode only)")
    sys.exit(0)
pb, eb, ov = ctx.column_totals(col)
print(f"{kind}: n={n} encode median {med:.3f} ms -> {n*8192/med/1e6:.1f} GB/s in; packed {pb/n:.0f} B/vec exc {eb/n:.0f} B/vec overflow {ov}")
# ... and the decode of what was just encoded (the ALP_RD column's decode has no other row in the round's profile)
out = torch.empty(n * 1024, dtype=torch.float64, device=dev)
dmed, _ = time_launches(lambda: ctx.decode(col, out), 5, 3)
print(f"{kind}: decode of the encoded column median {dmed:.3f} ms = {(n * 8192 + pb + eb + 13 * n) / dmed / 1e6 / 8000:.3f} of peak; round trip {bool(torch.equal(out.view(torch.int64), x.view(torch.int64)))}")
