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
