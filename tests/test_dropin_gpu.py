"""GPU: the source-compatible C++ header include/alp.hpp.  A C++ harness written in the style of the reference's own
unit test (test/test_alp_sample.cpp `test_column`) is compiled with g++ against include/ and libalpgpu.so and run over
the reference's test columns (tests/golden): every column must round-trip bit-exactly through
init -> encode -> analyze_ffor -> ffor -> falp -> patch_exceptions (or the ALP_RD chain), the fused and unfused decode
paths must agree, and the first vector must give the (bit width, exception count) the reference's test asserts."""
import os
import subprocess

import numpy as np
import pytest

import golden_io

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_dropin_header_against_reference_columns(tmp_path):
    exe = tmp_path / "dropin_test"
    subprocess.check_call(["g++", "-std=c++17", "-O1", f"-I{ROOT}/include", "-o", str(exe), f"{ROOT}/tests/cpp/dropin_test.cpp",
                           f"-L{ROOT}/alp_amd", "-lalpgpu", f"-Wl,-rpath,{ROOT}/alp_amd"])
    lines = []
    for name, col, gold, known in golden_io.first_vectors():
        key = name.replace("/", "_").replace(".csv", "")
        col.tofile(tmp_path / f"{key}.f64")
        is_rd = int(gold["scheme"][0] == 1)
        bw, exc = (int(known[0]), int(known[1])) if (known[0] >= 0 and not is_rd) else (-1, -1)
        lines.append(f"f64 {key} 1024 {bw} {exc} {is_rd}")
    for name, col, gold in golden_io.rowgroup_samples():  # 128 vectors: two rowgroups, k > 1 (second-level sampling)
        col.tofile(tmp_path / f"{name}.f64")
        lines.append(f"f64 {name} {col.size} {int(gold['bw'][0])} {int(gold['exc_cnt'][0])} {int(gold['scheme'][0] == 1)}")
    # a column that ends in a partial vector: the host-side sampler rules + whole vectors only
    tail = np.round(np.random.default_rng(1).uniform(0, 100, 3 * 1024 + 500), 2)
    tail.tofile(tmp_path / "partial_tail.f64")
    lines.append(f"f64 partial_tail {tail.size} -1 -1 0")
    # single precision: the reference's float test columns (asserted bit widths / exception count) and synthetic rowgroups
    for name, col, gold, known in golden_io.float_vectors():
        col.tofile(tmp_path / f"f32_{name}.f32")
        is_rd = int(gold["scheme"][0] == 1)
        bw = int(known[0]) if known[0] >= 0 else (int(gold["bw"][0]) if not is_rd else -1)
        exc = int(known[1]) if known[1] >= 0 else (int(gold["exc_cnt"][0]) if not is_rd else -1)
        lines.append(f"f32 f32_{name} {col.size} {bw} {exc} {is_rd}")
    tailf = np.round(np.random.default_rng(2).uniform(0, 100, 2 * 1024 + 333), 1).astype(np.float32)
    tailf.tofile(tmp_path / "partial_tail_f32.f32")
    lines.append(f"f32 partial_tail_f32 {tailf.size} -1 -1 0")
    (tmp_path / "columns.txt").write_text("\n".join(lines) + "\n")
    p = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True, timeout=600)
    tail_out = "\n".join(p.stdout.splitlines()[-15:])
    assert p.returncode == 0, f"drop-in harness failed:\n{tail_out}\n{p.stderr[-2000:]}"
    assert f"{len(lines)} columns, 0 failures" in p.stdout


def test_rowgroup_batched_cpp_surface_matches_the_per_vector_functions(tmp_path):
    """include/alp/batch.hpp: one call per rowgroup == the reference-shaped per-vector loop, byte for byte; prints both rates"""
    exe = tmp_path / "batch_test"
    subprocess.check_call(["g++", "-std=c++17", "-O1", f"-I{ROOT}/include", "-o", str(exe), f"{ROOT}/tests/cpp/batch_test.cpp",
                           f"-L{ROOT}/alp_amd", "-lalpgpu", f"-Wl,-rpath,{ROOT}/alp_amd"])
    p = subprocess.run([str(exe), "2"], capture_output=True, text=True, timeout=900)
    print(p.stdout)
    assert p.returncode == 0 and "batch_test: 0 failures" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]


def test_one_host_column_over_several_contexts(tmp_path):
    """include/alp/batch.hpp: alp::gpu::column<PT> over a device list = alpgpu_compress_host_multi_*: whole-rowgroup shards, one host thread and one
    pipeline per context, ONE blob — byte for byte the single-context blob.  Three contexts on this box's one GPU stand in for three GPUs."""
    exe = tmp_path / "multi_test"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", f"-I{ROOT}/include", "-o", str(exe), f"{ROOT}/tests/cpp/multi_test.cpp",
                           f"-L{ROOT}/alp_amd", "-lalpgpu", f"-Wl,-rpath,{ROOT}/alp_amd"])
    p = subprocess.run([str(exe), "0", "0", "0"], capture_output=True, text=True, timeout=900)
    print(p.stdout)
    assert p.returncode == 0 and "multi_test: 0 failures" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]
