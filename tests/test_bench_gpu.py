"""GPU: bench.py's N > 1 path (BASELINE.json configs[4]: ONE column sharded over the ranks, encode + decode under one clock) run end
to end with TWO ranks on the ONE GPU of the test box.  RCCL refuses two ranks on one device, so the test knob
ALPGPU_BENCH_TEST_SHARED_GPU puts both ranks on GPU 0 and lets them talk over gloo: the shard arithmetic, the rank > 0 code, the
reductions and rank 0's single JSON line are the driver's 8-GPU run in small; nothing measured here means anything."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_shard_one_column_and_rank0_prints_one_line():
    env = dict(os.environ, ALPGPU_BENCH_TEST_SHARED_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29641",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--column-gb", "2"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"rank 0 prints ONE JSON line, got {len(lines)}"
    r = json.loads(lines[0])
    assert p.stdout.rstrip().splitlines()[-1] == lines[0], "the JSON line is the last line of stdout"
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["steps"] == 3
    cfg = r["config"]
    n_total = int(2e9 / 8192) // 100 * 100
    assert cfg["column_vectors"] == n_total
    assert cfg["vectors_per_gpu"] * 2 >= n_total - 100, "two whole-rowgroup shards cover the column"
    assert r["value"] > 0 and r["roofline"]["frac"] > 0
    enc = r["encode"]
    assert enc["roundtrip_bit_exact_all_ranks"] is True and enc["overflow"] == 0 and enc["value"] > 0
    # every rank's own record: device identity, shard, decode and encode figures (BASELINE.json configs[4]: "per-GPU and aggregate")
    assert r["world_size_seen"] == 2 and [x["rank"] for x in r["ranks"]] == [0, 1]
    for x in r["ranks"]:
        assert x["vectors"] > 0 and x["first_vector"] % 100 == 0 and x["device"]
        assert x["decode_GBps"] > 0 and x["encode_GBps"] > 0 and x["decode_ms"] > 0 and x["encode_ms"] > 0 and x["roundtrip_bit_exact"] is True
    assert r["ranks"][1]["first_vector"] >= r["ranks"][0]["first_vector"] + r["ranks"][0]["vectors"], "rank 1's shard starts where rank 0's (possibly capped) shard ends or later"
    assert r["per_gpu_value_min"] <= r["per_gpu_value"]


def test_gpus_2_without_a_launcher_starts_its_own_ranks():
    """`python3 bench.py --gpus 2 --steps 3 --warmup 1 --column-gb 2` as typed — no torch.distributed.run around it: bench.py starts the
    two ranks itself (shared-GPU knob on this one-GPU box) and rank 0's line says n_gpus == 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(ALPGPU_BENCH_TEST_SHARED_GPU="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--column-gb", "2"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and p.stdout.rstrip().splitlines()[-1] == lines[0]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["encode"]["roundtrip_bit_exact_all_ranks"] is True
    # and a launcher whose world size disagrees with --gpus is refused before anything runs
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300,
                         env=dict(env, WORLD_SIZE="4", RANK="0"), cwd=ROOT)
    assert bad.returncode != 0 and "WORLD_SIZE=4" in bad.stderr
