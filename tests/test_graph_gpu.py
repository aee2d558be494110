"""Round 6: the whole-column entry points captured into a hipGraph and replayed (the library enqueues on the caller's stream and forks / joins its
second stream with events, so a stream capture around a call records all of it).  Every scenario runs in a process of its own: a capture that fails leaves the
process's stream in a state no later test should inherit.  Each prints RESULT <ok> <detail>; the bytes a replay writes must be the bytes the eager call writes."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HEAD = r"""
import sys, time
sys.path.insert(0, %r)
sys.path.insert(0, %r)
import numpy as np
import torch
import bench
import datagen
from alp_amd import capi
ctx = capi.Context(0)
side = torch.cuda.Stream()
def same(a, b):
    return bool(torch.equal(a.view(torch.uint8), b.view(torch.uint8)))
""" % (ROOT, os.path.join(ROOT, "tests"))

SCENARIOS = {
    # a hinted column of every packed width by rowgroup, 20 exceptions per vector: the plain store decode (one launch)
    "decode": r"""
col, _, _ = bench.build_decode_column(5300, 0, seed=3, exc_per_vec=20)
ref = ctx.decode(col).clone()
out = torch.zeros_like(ref)
with torch.cuda.stream(side):
    ctx.decode(col, out)          # warm-up on the capture stream
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        ctx.decode(col, out)
ok = True
for _ in range(3):
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    ok = ok and same(out, ref)
print("RESULT", ok, "replays of the store decode")
""",
    # the read-ahead forced on: the fork to the context's second stream and the join are part of the capture
    "decode_read_ahead": r"""
n = 40000
col, _, _ = bench.build_decode_column(n, 0, seed=5, bw_of_rowgroup=4, exc_per_vec=20)
ctx.set_option(capi.OPT_DECODE_READ_AHEAD, 0)
ref = ctx.decode(col).clone()
ctx.set_option(capi.OPT_DECODE_READ_AHEAD, 1)
assert ctx.decode_reads_ahead(col)
out = torch.zeros_like(ref)
with torch.cuda.stream(side):
    ctx.decode(col, out)
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        ctx.decode(col, out)
ok = True
for _ in range(4):
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    ok = ok and same(out, ref)
ctx.decode(col, out)              # and the context still works eagerly afterwards, read-ahead and all
torch.cuda.synchronize()
print("RESULT", ok and same(out, ref), "replays of the store decode with its read-ahead")
""",
    # the fused consumers
    "sum_and_count": r"""
col, _, _ = bench.build_decode_column(5300, 0, seed=4, exc_per_vec=20)
sums_ref = ctx.decode_sum(col).clone()
cnt_ref = ctx.decode_count_range(col, -1.0e3, 1.0e3).clone()
sums, cnt = torch.zeros_like(sums_ref), torch.zeros_like(cnt_ref)
with torch.cuda.stream(side):
    ctx.decode_sum(col, sums); ctx.decode_count_range(col, -1.0e3, 1.0e3, cnt)
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        ctx.decode_sum(col, sums)
        ctx.decode_count_range(col, -1.0e3, 1.0e3, cnt)
ok = True
for _ in range(3):
    sums.zero_(); cnt.zero_()
    g.replay()
    torch.cuda.synchronize()
    ok = ok and same(sums, sums_ref) and same(cnt, cnt_ref)
print("RESULT", ok, "replays of SUM and COUNT")
""",
    # encode + decode of a small column as ONE graph (the launch-bound case): the rowgroup search on the second stream, the vector encode, the gated recovery
    # kernels and the decode; replayed over new input in the same buffers
    "encode_decode": r"""
n = 700
xs = [torch.from_numpy(np.concatenate([datagen.mixed_column(400, seed=s), datagen.rd_column(200, seed=s + 1), datagen.drifting_column(100, seed=s + 2)])).cuda() for s in (11, 21, 31)]
x = xs[0].clone()
col = capi.DeviceColumn(n, 0)
out = torch.zeros_like(x)
eager = []
for xi in xs:                     # eager: the streams each input must give
    x.copy_(xi)
    ctx.encode(x, col)
    ctx.synchronize()
    eager.append([t.copy() for t in col.to_host()])
with torch.cuda.stream(side):
    ctx.encode(x, col); ctx.decode(col, out)
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        ctx.encode(x, col)
        ctx.decode(col, out)
ok = True
for xi, want in zip(xs, eager):
    x.copy_(xi)
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    ok = ok and same(out, xi)
    got = col.to_host()
    ok = ok and all(np.array_equal(a.view(np.uint8), b.view(np.uint8)) for a, b in zip(got, want))
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
t_graph = (time.perf_counter() - t0) / 20
t0 = time.perf_counter()
for _ in range(20):
    ctx.encode(x, col); ctx.decode(col, out)
torch.cuda.synchronize()
t_eager = (time.perf_counter() - t0) / 20
print("RESULT", ok, "encode+decode of %%d vectors: %%.0f us per replay, %%.0f us eager" %% (n, t_graph * 1e6, t_eager * 1e6))
""",
}


@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_entry_points_captured_into_a_graph_replay_the_same_bytes(name):
    p = subprocess.run([sys.executable, "-c", HEAD + SCENARIOS[name].replace("%%", "%")], capture_output=True, text=True, timeout=300)
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")]
    assert line, (name, p.stdout[-1500:], p.stderr[-3000:])
    print(line[-1])
    assert line[-1].split()[1] == "True", (name, line[-1])
