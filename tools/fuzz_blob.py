#!/usr/bin/env python3
"""fuzz_blob.py [rounds] [seed]: corrupted serialized columns.  A valid blob (ALP and ALP_RD rowgroups, exceptions, a ragged tail; double or float) gets 1-4 random
bytes overwritten — mostly in the header, the rowgroup states and the descriptors, sometimes anywhere — or is truncated, and goes through alpgpu_column_from_blob and
the chunked alpgpu_decompress_host_*.  Each must either REFUSE it (AlpGpuError) or accept it and then decode / sum it WITHOUT touching memory outside the column's
buffers (a wild access ends this process with a GPU memory fault, which is the failure this tool looks for).  What an accepted blob decodes to is
not checked — a flipped payload byte is a different column.  Progress is flushed every 200 rounds so that a fault can be placed."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import datagen  # noqa: E402
from alp_amd import capi  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
budget_s = float(os.environ.get("FUZZ_SECONDS", "1e9"))
ctx = capi.Context(0)


def valid_blob(f64, seed):
    if f64:
        data = np.concatenate([datagen.mixed_column(130, seed=seed, exc_rate=0.03), datagen.rd_column(101, seed=seed + 1), datagen.decimal_column(30, 3, seed=seed + 2)])
    else:
        data = np.concatenate([datagen.mixed_column_f32(130, seed=seed, exc_rate=0.03), datagen.rd_column_f32(101, seed=seed + 1), datagen.decimal_column_f32(30, 2, seed=seed + 2)])
    n_values = data.size - 391
    x = torch.from_numpy(data).cuda()
    ctx.pad_tail(x, n_values)
    col = ctx.encode(x)
    return ctx.to_blob(col, n_values), n_values


blobs = {True: valid_blob(True, 5), False: valid_blob(False, 9)}
refused = accepted = 0
t0 = time.time()
for r in range(rounds):
    if time.time() - t0 > budget_s:
        break
    rng = np.random.default_rng(91000 + seed0 * 100003 + r)
    f64 = bool(rng.integers(0, 2))
    blob, n_values = blobs[f64]
    hdr = np.frombuffer(blob[:64].tobytes(), np.uint64)
    n_vec, n_rg = int(hdr[3]), int(hdr[4])
    meta_end = 64 + 32 * n_rg + 32 * n_vec
    b = blob.copy()
    kind = int(rng.integers(0, 10))
    if kind == 0:
        b = b[: int(rng.integers(0, b.size))].copy()  # truncated
    else:
        for _ in range(int(rng.integers(1, 5))):
            pos = int(rng.integers(0, meta_end)) if kind < 8 else int(rng.integers(0, b.size))
            b[pos] = rng.integers(0, 256) if rng.integers(0, 2) else (int(b[pos]) ^ (1 << int(rng.integers(0, 8))))
    for route in (0, 1):
        try:
            if route == 0:
                col, nv = ctx.from_blob(b)
                out = ctx.decode(col)
                s = ctx.decode_sum(col)
                c = ctx.decode_count_range(col, -10.0, 10.0)
                ctx.synchronize()
                del out, s, c, col
            else:
                host = torch.empty((n_values + 1023) // 1024 * 1024 + 4096, dtype=torch.float64 if f64 else torch.float32)
                ctx.decompress_host(torch.from_numpy(b), host)
            accepted += 1
        except capi.AlpGpuError:
            refused += 1
        except (RuntimeError, MemoryError):  # the harness could not even allocate what the corrupted header asks for
            refused += 1
            torch.cuda.empty_cache()
    if r % 200 == 199:
        print(f"... {r + 1} rounds: {refused} refused, {accepted} accepted and decoded", flush=True)
print(f"fuzz_blob: {r + 1} rounds, {refused} refusals, {accepted} blobs accepted and decoded inside their buffers, {time.time() - t0:.0f} s", flush=True)
