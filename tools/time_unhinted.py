#!/usr/bin/env python3
"""time_unhinted.py [n_vectors]: what an UNHINTED decode costs (ALPGPU_OPT_DECODE_UNHINTED: sizes summed on the stream, the rule evaluated on the device, every
candidate shape launched gated): the hinted decode / the first unhinted decode (device plan, closed candidates) / the later ones (sizes learned) / the old
behaviour (option 0), per column; double and float.  Fractions of 8 TB/s; profiles/r06_decode_policy.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from alp_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
ctx = capi.Context(0)
out = torch.empty(n * 1024, dtype=torch.float64, device="cuda:0")
print(f"lib {bench.lib_sha16()}  n={n}")


def one(label, bw, exc, vb):
    c, _, ab = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=bw, exc_per_vec=exc, value_bytes=vb)
    o = out if vb == 8 else out.view(torch.float32)[: n * 1024]
    f = lambda ms: ab / ms / 1e6 / 8000  # noqa: E731
    hinted = f(bench.time_launches(lambda: ctx.decode(c, o), 7, 4)[0])
    hints = (int(c.c.packed_bytes_hint), int(c.c.exc_bytes_hint))
    c.c.packed_bytes_hint, c.c.exc_bytes_hint = 0, 0
    ctx.set_option(capi.OPT_DECODE_UNHINTED, 0)
    old = f(bench.time_launches(lambda: ctx.decode(c, o), 7, 4)[0])
    ctx.set_option(capi.OPT_DECODE_UNHINTED, 1)
    # first decodes: a fresh column object each time would be needed for "never seen"; force it by forgetting (an encode entry point forgets; here: toggle the key)
    firsts = []
    for _ in range(5):
        ctx.forget(c)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ctx.decode(c, o)
        b.record()
        torch.cuda.synchronize()
        firsts.append(a.elapsed_time(b))
    plan = ctx.unhinted_plan()
    firsts.sort()
    learned = f(bench.time_launches(lambda: ctx.decode(c, o), 7, 4)[0])
    c.c.packed_bytes_hint, c.c.exc_bytes_hint = hints
    print(f"{label:28s} hinted {hinted:.3f} | unhinted: first {f(firsts[2]):.3f} (shape {plan['shape']}, lead {plan['lead_max']}) later {learned:.3f} | option off {old:.3f}", flush=True)
    del c


for label, bw, exc in (("bw4", 4, 0), ("bw4+20exc", 4, 20), ("bw12", 12, 0), ("bw28", 28, 0), ("bw44", 44, 0), ("benchmark column", None, 0)):
    one("f64 " + label, bw, exc, 8)
for label, bw, exc in (("bw3", 3, 0), ("bw3+20exc", 3, 20), ("bw8", 8, 0), ("bw20", 20, 0), ("all widths", None, 0)):
    one("f32 " + label, bw, exc, 4)
