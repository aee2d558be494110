#!/usr/bin/env python3
"""r05_mall_warm.py: does a decode whose packed words / descriptors / exception records already sit in the 256 MiB Infinity Cache run faster, and can a bulk read
ahead of the decode ("pre-warm") put them there?  Measurement only (torch reductions stand in for a pre-warm kernel).

Part 1  small columns (N1 vectors, everything a launch reads fits the cache): per width / exceptions / vectors per workgroup / pad:
        warm  = launches back to back (what the launch before left in the cache),
        cold  = a 2 GiB read between launches (cache flushed),
        pre   = flush, then a bulk read of the column's streams, then the timed launch.
Part 2  1 Mi-vector columns decoded in segments of SEG_MB of input each through column views, the next segments' streams read on a side stream while
        a segment is decoded: whole = the library's one launch (auto rule), seg = segments without the reads, seg+pre = with them.
        (Part 2's seg+pre figures are CPU-bound — dozens of torch ops per segment — and were discarded; k_read_ahead replaced the idea.)
Fractions of 8 TB/s on algorithmic bytes."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from alp_amd import capi  # noqa: E402

dev = "cuda:0"
ctx = capi.Context(0)
flush_buf = torch.ones(1 << 29, dtype=torch.float32, device=dev)  # 2 GiB


def flush():
    return flush_buf.sum()


def bulk_read(col, vec, v0, v1):
    """read the streams of vectors [v0, v1): descriptors, packed words, exception records"""
    p0 = int(vec["packed_off"][v0])
    p1 = int(vec["packed_off"][v1 - 1]) + 128 * int(vec["bw"][v1 - 1])
    r = [col.vectors[v0 * 32: v1 * 32].view(torch.int64).sum(), col.packed[p0: (p1 + 7) // 8 * 8].view(torch.int64).sum()]
    if int(vec["exc_cnt"][v0:v1].max()) > 0:
        e0 = int(vec["exc_off"][v0])
        rec = (10 * int(vec["exc_cnt"].max()) + 7) // 8 * 8
        e1 = int(vec["exc_off"][v1 - 1]) + rec
        r.append(col.exc[e0:e1].view(torch.int64).sum())
    return r


def timed(fn, before, iters=7, warmup=3):
    for _ in range(warmup):
        before()
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        before()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def part1():
    n = int(os.environ.get("N1", str(128 << 10)))
    out = torch.empty(n * 1024, dtype=torch.float64, device=dev)
    widths = [int(w) for w in os.environ.get("WIDTHS1", "2,4,8,12").split(",")]
    print(f"part 1: {n} vectors, lib {bench.lib_sha16()}: bw exc vpw pad | warm cold pre", flush=True)
    for exc in (0, 20):
        for bw in widths:
            c, vec, ab = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=bw, exc_per_vec=exc)
            for vpw in (1, 2):
                for pad in (0, 14):
                    ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)
                    ctx.set_option(capi.OPT_DECODE_RESIDENCY_PAD, pad)
                    fr = lambda ms: ab / ms / 1e6 / 8000  # noqa: E731
                    warm = timed(lambda: ctx.decode(c, out), lambda: None)
                    cold = timed(lambda: ctx.decode(c, out), flush)
                    pre = timed(lambda: ctx.decode(c, out), lambda: (flush(), bulk_read(c, vec, 0, n)))
                    print(f"  {bw:>2} {exc:>2} vpw{vpw} pad{pad:>2} | {fr(warm):.3f} {fr(cold):.3f} {fr(pre):.3f}", flush=True)
            del c
    ctx.set_option(capi.OPT_DECODE_RESIDENCY_PAD, -1)
    ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)


def view_of(col, v0, v1):
    """struct alpgpu_column over vectors [v0, v1) of col (v0 on a rowgroup boundary): descriptors carry absolute stream offsets"""
    s = capi.CColumn()
    C.memmove(C.byref(s), C.byref(col.c), C.sizeof(capi.CColumn))
    s.n_vectors = v1 - v0
    s.n_rowgroups = (v1 - v0 + 99) // 100
    s.d_rowgroups = col.rowgroups.data_ptr() + (v0 // 100) * 32
    s.d_vectors = col.vectors.data_ptr() + v0 * 32
    frac = (v1 - v0) / col.n_vectors
    s.packed_bytes_hint = int(col.c.packed_bytes_hint * frac)
    s.exc_bytes_hint = int(col.c.exc_bytes_hint * frac)
    return s


def part2():
    n = 1 << 20
    out = torch.empty(n * 1024, dtype=torch.float64, device=dev)
    seg_mb = [int(x) for x in os.environ.get("SEG_MB", "64,128").split(",")]
    cases = [(bw, exc) for exc in (0, 20) for bw in [int(w) for w in os.environ.get("WIDTHS2", "2,4,8,16,28").split(",")]] + [(None, 0)]
    side = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream()
    print(f"part 2: {n} vectors: bw exc | whole(auto vpw) | per SEG_MB {seg_mb}: [vpw1: seg seg+pre | vpw2: seg seg+pre]", flush=True)
    for bw, exc in cases:
        c, vec, ab = bench.build_decode_column(n, 0, seed=7, bw_of_rowgroup=bw, exc_per_vec=exc)
        fr = lambda ms: ab / ms / 1e6 / 8000  # noqa: E731
        ctx.set_option(capi.OPT_DECODE_RESIDENCY_PAD, -1)
        ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)
        whole = timed(lambda: ctx.decode(c, out), lambda: None, 5, 3)
        auto_vpw = ctx.decode_vectors_per_wg(c)
        line = f"  {str(bw):>4} {exc:>2} | {fr(whole):.3f} ({auto_vpw}) |"
        per_vec = (128 * vec["bw"].astype(np.int64) + 32 + 10 * vec["exc_cnt"].astype(np.int64))
        cum = np.cumsum(per_vec)
        for mb in seg_mb:
            # segment ends: rowgroup boundaries where the input read so far crosses multiples of mb MiB
            ends, last = [], 0
            target = mb << 20
            while last < n:
                base = cum[last - 1] if last else 0
                e = int(np.searchsorted(cum, base + target, side="left")) + 1
                e = min(n, max(last + 100, (e // 100) * 100))
                ends.append(e)
                last = e
            segs = list(zip([0] + ends[:-1], ends))
            views = [view_of(c, a, b) for a, b in segs]
            outs = [out.data_ptr() + a * 8192 for a, _ in segs]
            line += f" {mb}MiB x{len(segs)}:"
            for vpw in (1, 2):
                ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, vpw)

                def run(pre):
                    evs = {}
                    if pre:
                        side.wait_stream(main)
                        with torch.cuda.stream(side):
                            for k in range(min(2, len(segs))):
                                bulk_read(c, vec, *segs[k])
                                evs[k] = torch.cuda.Event()
                                evs[k].record(side)
                    for k in range(len(segs)):
                        if pre:
                            main.wait_event(evs[k])
                        ctx._call("decode", "f64", C.byref(views[k]), C.c_void_p(outs[k]))
                        if pre and k + 2 < len(segs):
                            done = torch.cuda.Event()
                            done.record(main)
                            with torch.cuda.stream(side):
                                side.wait_event(done)
                                bulk_read(c, vec, *segs[k + 2])
                                evs[k + 2] = torch.cuda.Event()
                                evs[k + 2].record(side)
                    if pre:
                        main.wait_stream(side)

                a = timed(lambda: run(False), lambda: None, 5, 2)
                b = timed(lambda: run(True), lambda: None, 5, 2)
                line += f" [{fr(a):.3f} {fr(b):.3f}]"
        print(line, flush=True)
        del c
    ctx.set_option(capi.OPT_DECODE_VECTORS_PER_WG, 0)


if __name__ == "__main__":
    which = os.environ.get("PARTS", "1,2").split(",")
    if "1" in which:
        part1()
    if "2" in which:
        part2()
