#!/usr/bin/env python3
"""debug_consume.py: decode_sum of small columns, repeated, against the documented order (which vectors differ, how often)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import datagen, layout
from alp_amd import capi
from oracle.pyoracle import Oracle
from test_decode_sum_gpu import host_sums_pipelined as host_sums, _same_bits
o = Oracle(); ctx = capi.Context(0); ctx.set_option(capi.OPT_CONSUMER_PIPELINED, 1)
tag = os.path.basename(os.environ.get("ALPGPU_LIB", "libalpgpu.so"))
for name, col in (("decimal2", datagen.decimal_column(130, 2, seed=1)), ("noexc", np.round(np.random.default_rng(1).uniform(-1000, 1000, 130 * 1024), 2)),
                  ("long", datagen.decimal_column(6000, 2, seed=3))):
    enc = o.encode_column(col)
    dcol = capi.DeviceColumn.from_host(*layout.compact(enc))
    want = host_sums(col.reshape(-1, 1024))
    bad = {}
    for rep in range(20):
        got = ctx.decode_sum(dcol).cpu().numpy()
        for v in np.nonzero(~_same_bits(got, want))[0]:
            bad[int(v)] = bad.get(int(v), 0) + 1
    if os.environ.get("VERIFY"):
        got = ctx.decode_sum(dcol).cpu().numpy()
        nz = np.nonzero(got)[0]
        print(tag, name, "ring mismatches (vector: units + 1000 * (first bad unit + 1)), bw", int(enc["bw"][0]), {int(v): int(got[v]) for v in nz[:30]}, "vectors with mismatches:", nz.size, flush=True)
        continue
    print(tag, name, "max exc", int(enc["exc_cnt"].max()), "bad vectors (index: times of 20):", dict(sorted(bad.items())[:40]), "n_bad", len(bad), flush=True)
