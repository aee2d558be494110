"""GPU: alpgpu_debug_traffic_probe — the measurement aid bench.py times next to the encode legs (include/alpgpu.h).  It is not part
of the codec, but it is part of the ABI: its output is a pure function of its input (sum of the vector's sixteen-byte units, plus
the vector index and the unit index), it writes exactly write_bytes per vector, and it rejects sizes it cannot honour."""
import numpy as np
import pytest
import torch

from alp_amd import capi

pytestmark = pytest.mark.gpu


def test_probe_output_is_the_documented_function_of_its_input():
    dev = torch.device("cuda:0")
    ctx = capi.Context(0)
    n, wb = 37, 4336  # an odd number of vectors (partial last tile), the mixed column's bytes per vector
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    x = torch.randint(-(1 << 40), 1 << 40, (n * 1024,), dtype=torch.int64, device=dev, generator=g)
    out = torch.full((n * wb // 8 + 16,), -1, dtype=torch.int64, device=dev)
    ctx.traffic_probe(x, out, n, wb)
    ctx.synchronize()
    got = out.cpu().numpy()
    xs = x.cpu().numpy().view(np.uint64).reshape(n, 8, 64, 2)  # vector, step, lane, element of the 16-byte unit
    lane_sum = xs.sum(axis=1, dtype=np.uint64)                   # what a lane has added up: [n, 64, 2]
    units = wb // 16
    for v in range(n):
        u = np.arange(units)
        exp_x = lane_sum[v, u % 64, 0] + np.uint64(v) + u.astype(np.uint64)
        exp_y = lane_sum[v, u % 64, 1] + np.uint64(1)
        rec = got[v * units * 2: (v + 1) * units * 2].view(np.uint64).reshape(units, 2)
        assert np.array_equal(rec[:, 0], exp_x) and np.array_equal(rec[:, 1], exp_y), f"vector {v}"
    assert np.all(got[n * units * 2:] == -1), "nothing is written past the last vector's bytes"


def test_probe_rejects_what_it_cannot_do():
    dev = torch.device("cuda:0")
    ctx = capi.Context(0)
    x = torch.zeros(1024, dtype=torch.float64, device=dev)
    out = torch.zeros(2048, dtype=torch.float64, device=dev)
    for bad in (8, 4330, 8208):
        with pytest.raises(capi.AlpGpuError):
            ctx.traffic_probe(x, out, 1, bad)
    ctx.traffic_probe(x, out, 0, 4336)  # an empty column is fine
    ctx.traffic_probe(x, out, 1, 0)     # and so is writing nothing
    ctx.synchronize()


def test_decode_probe_runs_on_a_real_column_and_touches_only_its_output():
    import datagen
    dev = torch.device("cuda:0")
    ctx = capi.Context(0)
    x = torch.from_numpy(datagen.mixed_column(230, seed=11, exc_rate=0.02)).to(dev)
    col = ctx.encode(x)
    out = torch.full((230 + 8,), -7.0, dtype=torch.float64, device=dev)
    ctx.decode_probe(col, out)
    ctx.synchronize()
    got = out.cpu().numpy()
    assert np.all(got[230:] == -7.0), "one double per vector, nothing behind them"
    assert np.all((got[:230] >= 0) & (got[:230] == np.floor(got[:230]))), "the probe's output is a small non-negative integer per vector"
    back = ctx.decode(col)
    ctx.synchronize()
    assert torch.equal(back.view(torch.int64), x.view(torch.int64)), "the column is untouched"
