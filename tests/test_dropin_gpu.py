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


def test_per_vector_api_from_several_host_threads(tmp_path):
    """The reference's vector functions are re-entrant (its scratch is thread_local since issue #41, encoder.hpp:314-319; its end-to-end benchmark decodes from
    worker threads, run_query.cpp:233-305).  Six host threads run the reference-shaped loop over columns of their own (double and float; decimal, ALP_RD,
    exceptions, specials) through include/alp.hpp at the same time — one process-wide context and stream, device scratch per thread (gpu_bridge.hpp): every byte
    they produce equals what the same loop produced alone, and every vector round-trips."""
    exe = tmp_path / "threads_test"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", f"-I{ROOT}/include", "-o", str(exe), f"{ROOT}/tests/cpp/threads_test.cpp",
                           f"-L{ROOT}/alp_amd", "-lalpgpu", f"-Wl,-rpath,{ROOT}/alp_amd"])
    p = subprocess.run([str(exe), "6", "130", "2"], capture_output=True, text=True, timeout=900)
    print(p.stdout)
    assert p.returncode == 0 and ", 0 failures" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]


def test_header_tables_are_the_devices_and_encode_value(tmp_path, ctx, oracle):
    """alp::Constants<PT>::{FRAC_ARR, EXP_ARR, FACT_ARR} (host copies in include/alp/constants.hpp) against what the DEVICE computes with ITS tables,
    and alp::encoder<PT>::encode_value<SAFE> (through alpgpu_encode_value_*) against the oracle."""
    import ctypes
    import torch
    exe = tmp_path / "constants_dump"
    subprocess.check_call(["g++", "-std=c++17", "-O1", f"-I{ROOT}/include", "-o", str(exe), f"{ROOT}/tests/cpp/constants_dump.cpp", f"-L{ROOT}/alp_amd", "-lalpgpu",
                           f"-Wl,-rpath,{ROOT}/alp_amd"])
    probes = [("ev64", v, f, e) for v, f, e in ((123.45, 0, 2), (-0.0, 0, 0), (1e300, 0, 18), (92233.72, 12, 14), (float("nan"), 1, 3), (-7.125, 3, 3), (9.3e18, 0, 0))]
    args = [str(a) for p in probes for a in p]
    out = subprocess.run([str(exe), *args], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    tab = {}
    ev = []
    for line in out.stdout.splitlines():
        w = line.split()
        if w[0].startswith("ev"):
            ev.append((int(w[1]), int(w[2])))
        else:
            tab.setdefault(w[0], []).append(int(w[2], 16))
    assert [len(tab[k]) for k in ("frac64", "exp64", "fact64", "frac32", "exp32", "fact32")] == [21, 24, 19, 11, 11, 10]
    # double tables through the device: decode_values(1, fac, exp = 0) = (double)FACT[fac] * FRAC[0]; decode_values(1, 0, exp) = FRAC[exp];
    # encode_value(1.0, 0, exp) = the integer EXP[exp] (exact for exp <= 18)
    one = torch.ones(19, 1024, dtype=torch.int64, device="cuda")
    outd = torch.empty(19, 1024, dtype=torch.float64, device="cuda")
    idx = torch.arange(19, dtype=torch.uint8, device="cuda")
    zero = torch.zeros(19, dtype=torch.uint8, device="cuda")
    ctx.decode_values(one, outd, zero, idx)
    ctx.synchronize()
    assert [int(x) for x in outd[:, 0].cpu().numpy().view(np.uint64)] == tab["frac64"][:19]
    ctx.decode_values(one, outd, idx, zero)
    ctx.synchronize()
    assert [int(x) for x in outd[:, 0].cpu().numpy()] == [int(np.array(b, dtype=np.uint64).view(np.int64)) for b in tab["fact64"]]
    ones = torch.ones(4, dtype=torch.float64, device="cuda")
    for e in range(19):
        got = int(ctx.encode_value(ones, 0, e, safe=False)[0])
        assert got == int(np.array(tab["exp64"][e], dtype=np.uint64).view(np.float64)) == 10 ** e
    # the float tables and the rest of the double ones: correctly rounded powers of ten, as the reference's literals are
    assert tab["frac64"] == [int(np.float64(f"1e-{i}").view(np.uint64)) for i in range(21)] and tab["exp64"] == [int(np.float64(f"1e{i}").view(np.uint64)) for i in range(24)]
    assert tab["frac32"] == [int(np.float32(f"1e-{i}").view(np.uint32)) for i in range(11)] and tab["exp32"] == [int(np.float32(f"1e{i}").view(np.uint32)) for i in range(11)]
    assert tab["fact32"] == [10 ** i for i in range(10)]
    # encode_value<SAFE> / <!SAFE> of the header against the oracle
    lib = oracle.lib
    lib.alpo_encode_value_safe.restype = lib.alpo_encode_value_unsafe.restype = ctypes.c_int64
    for (_, v, f, e), (safe, unsafe) in zip(probes, ev):
        assert safe == lib.alpo_encode_value_safe(ctypes.c_double(v), f, e) and unsafe == lib.alpo_encode_value_unsafe(ctypes.c_double(v), f, e), (v, f, e)
    # and a batch through the C ABI, boundary values included
    import datagen
    x = np.concatenate([datagen.search_boundary_values(), datagen.decimal_column(2, 3, seed=4)[:500]]) if hasattr(datagen, "search_boundary_values") else datagen.decimal_column(2, 3, seed=4)
    xd = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).cuda()
    for f, e in ((0, 0), (2, 5), (12, 14), (0, 18)):
        gs, gu = ctx.encode_value(xd, f, e, safe=True).cpu().numpy(), ctx.encode_value(xd, f, e, safe=False).cpu().numpy()
        ws = np.array([lib.alpo_encode_value_safe(ctypes.c_double(float(v)), f, e) for v in x], dtype=np.int64)
        wu = np.array([lib.alpo_encode_value_unsafe(ctypes.c_double(float(v)), f, e) for v in x], dtype=np.int64)
        assert np.array_equal(gs, ws) and np.array_equal(gu, wu), (f, e)
