#!/usr/bin/env python3
"""fuzz_host.py [rounds] [seed]: the host-column entry points (alpgpu_compress_host_* / alpgpu_decompress_host_* and their _multi forms) over random LENGTHS — ragged
tails, lengths around the pipeline's chunk (12 800 vectors) and rowgroup boundaries, empty and one-value columns — random contents (tests/test_fuzz_gpu.py's generator),
both precisions, pinned and pageable memory, 1-4 contexts on the one GPU.  Per round: the blob of one context = the blob of k contexts, byte for byte; decompressed by
one and by j contexts it is the input, bit for bit, tail included.  FUZZ_SECONDS bounds the run."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from alp_amd import capi  # noqa: E402
from test_fuzz_gpu import fuzz_column  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
budget_s = float(os.environ.get("FUZZ_SECONDS", "1e9"))
ctxs = [capi.Context(0) for _ in range(4)]
CHUNK = 12800
fails, done, t0 = 0, 0, time.time()
for r in range(rounds):
    if time.time() - t0 > budget_s:
        break
    rng = np.random.default_rng(77000 + seed0 * 100003 + r)
    f64 = bool(rng.integers(0, 2))
    kind = int(rng.integers(0, 6))
    if kind == 0:
        n_values = int(rng.integers(0, 3000))
    elif kind == 1:  # around a chunk boundary of the pipeline
        n_values = int(rng.integers(1, 4)) * CHUNK * 1024 + int(rng.integers(-2100, 2100))
    elif kind == 2:  # around a rowgroup boundary
        n_values = int(rng.integers(1, 60)) * 102400 + int(rng.integers(-1100, 1100))
    else:
        n_values = int(np.exp(rng.uniform(np.log(1024.0), np.log(30.0e6))))
    base = fuzz_column(rng, np.float64 if f64 else np.float32)
    col_np = np.resize(base, n_values) if n_values else base[:0]
    x = torch.from_numpy(np.ascontiguousarray(col_np))
    if rng.integers(0, 2) and n_values:
        xp = torch.empty(n_values, dtype=x.dtype, pin_memory=True)
        xp.copy_(x)
        x = xp
    k, j = int(rng.integers(1, 5)), int(rng.integers(1, 5))
    what = f"round {r} seed {seed0} {'f64' if f64 else 'f32'} n_values={n_values} compress x{k} decompress x{j} pinned={x.is_pinned()}"
    try:
        blob1 = ctxs[0].compress_host(x)
        blobk = capi.Context.compress_host_multi(ctxs[:k], x)
        if not torch.equal(blob1, blobk):
            fails += 1
            print(f"FAIL {what}: the blob of {k} contexts differs from the one-context blob ({blob1.numel()} / {blobk.numel()} bytes)")
        it = torch.int64 if f64 else torch.int32
        n_pad = (n_values + 1023) // 1024 * 1024
        for name, fn in (("one context", lambda o: ctxs[1].decompress_host(blob1, o)), (f"{j} contexts", lambda o: capi.Context.decompress_host_multi(ctxs[:j], blobk, o))):
            exact = bool(rng.integers(0, 2))  # a buffer of exactly n_values values, or of whole vectors: nothing may be written behind either
            cap = n_values if exact else n_pad
            buf = torch.full((cap + 64,), -7.0, dtype=x.dtype)
            out = buf[:cap]
            nv = int(fn(out))
            if nv != n_values or not torch.equal(out[:n_values].view(it), x.view(it)):
                fails += 1
                print(f"FAIL {what}: decompressed by {name} (capacity {cap}): {nv} values, bytes {'differ' if nv == n_values else 'n/a'}")
            if not bool((buf[cap:] == -7.0).all()):
                fails += 1
                print(f"FAIL {what}: decompressed by {name}: wrote behind its buffer of {cap} values")
    except Exception as exc:  # noqa: BLE001
        fails += 1
        print(f"FAIL {what}: {type(exc).__name__}: {exc}")
    done += 1
print(f"fuzz_host: {done} rounds, {fails} failures, {time.time() - t0:.0f} s", flush=True)
sys.exit(1 if fails else 0)
