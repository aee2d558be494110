"""Round 6: the whole-column C ABI from several host threads at once, one context and one stream per thread on the same device (the shape of the reference's
morsel-driven workers, publication/source_code/bench_end_to_end/src/benchmarks/alp/run_query.cpp:233-305: each worker owns its buffers and calls the codec on its
own).  ctypes releases the interpreter lock during the calls, so the library's entry points really overlap.  Every thread's streams, decoded bytes, sums and counts
must be what the same calls give alone."""
import os
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_contexts_of_several_threads_encode_and_decode_side_by_side():
    import torch
    import datagen
    from alp_amd import capi

    inputs = [
        np.concatenate([datagen.mixed_column(900, seed=3), datagen.rd_column(300, seed=4)]),
        datagen.rd_column(1100, seed=5),
        datagen.drifting_column(1300, seed=6),
        np.concatenate([datagen.mixed_column_f32(800, seed=7), datagen.rd_column_f32(300, seed=8)]),
        datagen.decimal_column(1000, decimals=1, seed=9),
    ]

    def work(x, ctx, rounds):
        res = []
        for _ in range(rounds):
            col = ctx.encode(x)
            ctx.column_totals(col)
            out = ctx.decode(col)
            sums = ctx.decode_sum(col)
            cnt = ctx.decode_count_range(col, -50.0, 50.0)
            ctx.synchronize()
            res.append([a.copy() for a in col.to_host()] + [out.cpu().numpy().view(np.uint8), sums.cpu().numpy().view(np.uint8), cnt.cpu().numpy()])
        return res

    xs = [torch.from_numpy(a).cuda() for a in inputs]
    alone_ctx = capi.Context(0)
    alone = [work(x, alone_ctx, 1)[0] for x in xs]
    for a, x in zip(alone, inputs):  # (and the decode is the input, bit for bit)
        assert np.array_equal(a[4], x.view(np.uint8))

    got, errors = [None] * len(xs), []

    def thread_main(i):
        try:
            ctx = capi.Context(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                got[i] = work(xs[i], ctx, 3)
        except Exception as exc:  # noqa: BLE001 (reported below)
            errors.append((i, repr(exc)))

    torch.cuda.synchronize()
    threads = [threading.Thread(target=thread_main, args=(i,)) for i in range(len(xs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i, rounds in enumerate(got):
        for r, res in enumerate(rounds):
            for k, (a, b) in enumerate(zip(res, alone[i])):
                assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), (i, r, k)
