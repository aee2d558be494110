"""GPU: include/alpgpu.h from plain C.  tests/c/abi_roundtrip.c is compiled with gcc -std=c11 against include/ and
libalpgpu.so (no C++, no HIP headers on the caller's side) and run: encode + decode of doubles and floats resident in HBM,
totals, blob round trip, error reporting."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_from_plain_c(tmp_path):
    exe = tmp_path / "abi_roundtrip"
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-Werror", f"-I{ROOT}/include", "-o", str(exe),
                           f"{ROOT}/tests/c/abi_roundtrip.c", f"-L{ROOT}/alp_amd", "-lalpgpu", "-lm", f"-Wl,-rpath,{ROOT}/alp_amd"])
    p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "ok   f64" in p.stdout and "ok   f32" in p.stdout and "0 failures" in p.stdout
