"""CPU: the parts of bench.py that decide what the JSON line may claim, without a GPU — roofline.traffic is quoted only from a
profile taken with the very library that is loaded; the algorithmic byte counts follow SURVEY.md §8(d); the on-device column
generators are only called on the GPU and are not covered here."""
import hashlib
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def bench():
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    return importlib.import_module("bench")


def test_traffic_is_quoted_only_for_the_loaded_library(bench, tmp_path, monkeypatch):
    real = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    assert set(("kernel", "vectors", "hbm_bytes_per_launch", "lib_sha16", "source")) <= set(real)
    # a scratch repo root with a profile whose hash is / is not the library's
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "lib_sha16", lambda: "0123456789abcdef")
    for sha, quoted in (("0123456789abcdef", True), ("fedcba9876543210", False)):
        json.dump({"kernel": "k_decode_column", "vectors": 1 << 20, "hbm_bytes_per_launch": 123, "lib_sha16": sha, "source": "test"},
                  open(tmp_path / "profiles" / "hbm_traffic.json", "w"))
        r = {"roofline": {"traffic": None}}
        bench.traffic_from_profile(r, 1 << 20)
        assert (r["roofline"]["traffic"] == 123) is quoted
        assert ("traffic_note" in r["roofline"]) is (not quoted)
    r = {"roofline": {"traffic": None}}
    bench.traffic_from_profile(r, 1 << 19)  # another column size: not this profile's launch
    assert r["roofline"]["traffic"] is None


def test_lib_sha16_is_the_hash_of_the_library_file(bench):
    from alp_amd import capi
    path = capi.LIB_PATH if hasattr(capi, "LIB_PATH") else os.path.join(ROOT, "alp_amd", "libalpgpu.so")
    want = hashlib.sha256(open(os.environ.get("ALPGPU_LIB", path), "rb").read()).hexdigest()[:16]
    assert bench.lib_sha16() == want


def test_algorithmic_bytes_follow_the_survey(bench):
    # encode: read 8192 per vector once, write packed + exception bytes + 13 B of metadata per vector
    assert bench.encode_alg_bytes(10, 1000, 200) == 10 * 8192 + 1000 + 200 + 130


# ---- `python bench.py --gpus N` without a launcher starts its own ranks; a world size that disagrees with --gpus is an error ------------
def test_launch_plan(bench):
    assert bench.launch_plan(1, {}, 0) == ("run",)
    assert bench.launch_plan(1, {"WORLD_SIZE": "1"}, 8) == ("run",)
    assert bench.launch_plan(8, {"WORLD_SIZE": "8"}, 8) == ("run",)              # the driver's torch.distributed.run form
    assert bench.launch_plan(8, {}, 8) == ("spawn", 8)                           # the N = 1 form typed with another N
    assert bench.launch_plan(2, {"ALPGPU_BENCH_TEST_SHARED_GPU": "1"}, 1) == ("spawn", 2)
    for gpus, env, visible in ((8, {"WORLD_SIZE": "4"}, 8), (1, {"WORLD_SIZE": "8"}, 8), (8, {}, 1), (0, {}, 8)):
        assert bench.launch_plan(gpus, env, visible)[0] == "error"


def _run_bench(argv, env_extra, timeout=600):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_gpus_2_without_a_launcher_runs_two_ranks_and_prints_one_line():
    """the exact command of the GPU twin (tests/test_bench_gpu.py), here with the dry-run knob: launcher, gloo rendezvous on 127.0.0.1,
    barrier, reductions, ONE line from rank 0 with n_gpus == 2"""
    p = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--column-gb", "2"], {"ALPGPU_BENCH_DRY_RUN": "1"})
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and p.stdout.rstrip().splitlines()[-1] == lines[0]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["warmup"] == 1
    assert r["config"]["vectors_covered_by_the_shards"] == r["config"]["column_vectors"]


def test_world_size_that_disagrees_with_gpus_exits_non_zero():
    p = _run_bench(["--gpus", "2"], {"WORLD_SIZE": "4", "RANK": "0", "ALPGPU_BENCH_DRY_RUN": "1"}, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=4" in p.stderr and not [l for l in p.stdout.splitlines() if l.startswith("{")]
    p = _run_bench(["--gpus", "8"], {}, timeout=120)  # no launcher, and (here) no 8 GPUs either
    assert p.returncode != 0 and "GPU(s) are visible" in p.stderr


def test_gpus_4_dry_run_returns_four_rank_records():
    """the one 8-GPU shot must come back with EVERY rank's figures: rank 0's line carries `ranks` (one record per rank, gathered through the
    process group) and the world size the group itself reports"""
    p = _run_bench(["--gpus", "4", "--steps", "2", "--warmup", "1", "--column-gb", "2"], {"ALPGPU_BENCH_DRY_RUN": "1"})
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert r["n_gpus"] == 4 and r["world_size_seen"] == 4
    assert [x["rank"] for x in r["ranks"]] == [0, 1, 2, 3]
    assert sum(x["vectors"] for x in r["ranks"]) == r["config"]["column_vectors"]
    firsts = [x["first_vector"] for x in r["ranks"]]
    assert firsts == sorted(firsts) and firsts[0] == 0 and all(f % 100 == 0 for f in firsts), "contiguous whole-rowgroup shards in rank order"
    assert all("device" in x for x in r["ranks"])
