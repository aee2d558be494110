#!/usr/bin/env python3
"""Timing of the float path on one GPU: encode (init + vectors) and decode at each launch shape, HIP events."""
import json
import sys
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alp_amd import capi  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    ctx = capi.Context(0)
    res = {}
    for name, gen in {
        "decimal2_0_1000": lambda: (torch.round(torch.rand(n * 1024, device="cuda", dtype=torch.float64) * 1e5) / 100).to(torch.float32),
        "decimal1_mixed_exc1pct": lambda: torch.where(torch.rand(n * 1024, device="cuda") < 0.01,
                                                      torch.rand(n * 1024, device="cuda") * 3.14159,
                                                      (torch.round(torch.rand(n * 1024, device="cuda", dtype=torch.float64) * 1e4) / 10).to(torch.float32)),
        "rd_unit": lambda: torch.rand(n * 1024, device="cuda", dtype=torch.float32),
    }.items():
        x = gen().contiguous()
        col = capi.DeviceColumn(n, dtype="f32")
        t_enc = timed(lambda: ctx.encode(x, col), 3)
        pb, eb, ov = ctx.column_totals(col)
        out = torch.empty_like(x)
        r = {"n_vectors": n, "packed_bytes": pb, "exc_bytes": eb, "encode_ms": t_enc, "encode_GBps_in": n * 4096 / t_enc / 1e6}
        for vpw in (1, 2, 4):
            ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
            t = timed(lambda: ctx.decode(col, out))
            r[f"decode_ms_v{vpw}"] = t
            r[f"decode_TBps_total_v{vpw}"] = (n * (4096 + 32) + pb + eb) / t / 1e9
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)
        assert torch.equal(out.view(torch.int32), x.view(torch.int32)), name
        res[name] = r
        del x, col, out
        torch.cuda.empty_cache()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
