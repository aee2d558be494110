"""Rules a build of the kernels has to meet that no compiler checks (CPU only: hipcc cross-compiles to assembly here)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None, reason="no hipcc")
def test_no_kernel_reads_the_last_register_of_its_allocation_in_a_64_bit_instruction(capsys):
    """gfx950 range-checks a single-register operand of a 64-bit instruction (the shift amount of v_lshrrev_b64 ...) as a register pair: in the
    last register of the allocation it is 'out of range' and VGPR0 is read instead (tools/last_vgpr_probe.hip).  Builds of k_sink_direct
    with that pattern returned wrong sums for ~5 % of some vectors (profiles/r03_consumers.txt); the allocator does not know the rule."""
    import check_top_vgpr
    rc = check_top_vgpr.main()
    out = capsys.readouterr().out
    assert rc == 0, out


def test_the_checker_sees_the_pattern():
    import check_top_vgpr
    asm = """
_ZN1a1kEv:
	v_lshlrev_b64 v[30:31], v63, v[30:31]
	s_endpgm
	.amdhsa_next_free_vgpr 64
_ZN1a1jEv:
	v_lshlrev_b64 v[30:31], v62, v[30:31]
	v_add_u32_e32 v1, v63, v2
	s_endpgm
	.amdhsa_next_free_vgpr 64
_ZN1a1iEv:
	v_lshrrev_b64 v[4:5], v64, v[4:5]
	s_endpgm
	.amdhsa_next_free_vgpr 65
"""
    found = check_top_vgpr.kernels_with_pattern(asm)
    assert [(n, k) for n, k, _ in found] == [("_ZN1a1kEv", 64)]


def test_committed_profiles_are_quoted_only_for_the_built_library():
    """profiles/hbm_traffic.json (what bench.py quotes as roofline.traffic) carries the sha-256 prefix of the library it was measured with.  Either
    the library in the tree is that one (builds are deterministic: same sources, same compiler, same bytes), or bench.py must NOT quote the
    file (a rebuilt library whose profile has not been retaken yet): there is no third state in which a stale number gets printed."""
    import hashlib
    import json
    lib = os.path.join(ROOT, "alp_amd", "libalpgpu.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    sys.path.insert(0, ROOT)
    import bench
    sha = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]
    prof = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    r = {"roofline": {"traffic": None}}
    bench.traffic_from_profile(r, prof["vectors"])
    if prof["lib_sha16"] == sha and not os.environ.get("ALPGPU_LIB"):
        assert r["roofline"]["traffic"] == prof["hbm_bytes_per_launch"]
    else:
        assert r["roofline"]["traffic"] is None and "traffic_note" in r["roofline"]


def test_every_entry_point_makes_its_contexts_device_current(capsys):
    """hipSetDevice is per host thread: a process that drives the GPUs of a node from several threads (alpgpu_compress_host_multi_*, or a caller's
    own threads) must get the context's device on whatever thread it calls from — tools/audit_set_device.py walks include/alpgpu.h's entry points
    in alp_amd/csrc/api_*.hip (the in-process multi-GPU path has never met a second physical GPU)"""
    import audit_set_device
    rc = audit_set_device.main()
    out = capsys.readouterr().out
    assert rc == 0, out


def test_the_built_library_itself_is_checked(capsys):
    """the same rule on the ARTEFACT (the assembly check recompiles the sources with its own flag list, which can drift from the real
    build): tools/check_top_vgpr.py --library unbundles the code objects of alp_amd/libalpgpu.so, disassembles them and reads each kernel's
    allocation (granule and accum offset) from its descriptor"""
    import check_top_vgpr
    lib = os.path.join(ROOT, "alp_amd", "libalpgpu.so")
    if not os.path.exists(lib) or not os.path.exists(os.path.join(check_top_vgpr.LLVM, "llvm-objdump")):
        pytest.skip("library or llvm-objdump missing")
    rc = check_top_vgpr.check_library(lib)
    out = capsys.readouterr().out
    assert rc == 0, out
    assert len(check_top_vgpr.library_kernels(lib)) >= 70


def test_an_instruction_no_probe_has_cleared_fails_the_check():
    """an instruction that reads the top register beside a 64-bit operand and is in neither list (convicted by the probes / cleared by them) is
    reported, so that a new compiler idiom cannot slip through unprobed"""
    import check_top_vgpr
    body = ["v_lshrrev_b64 v[2:3], v63, v[2:3]", "v_ldexp_f64 v[2:3], v[4:5], v63", "ds_read_b64 v[4:5], v63", "v_new_thing_b64 v[4:5], v63, v[6:7]",
            "v_mul_f64 v[0:1], v[62:63], v[2:3]", "v_add_u32_e32 v1, v63, v2"]
    convicted, unprobed = check_top_vgpr.top_register_hits(body, "v63")
    assert convicted == ["v_lshrrev_b64 v[2:3], v63, v[2:3]"] and unprobed == ["v_new_thing_b64 v[4:5], v63, v[6:7]"]


def _kernel_args_from_library(lib):
    """kernel name -> [(offset, size, value_kind)] from the AMDGPU metadata note (msgpack) of every gfx950 code object in the library"""
    import subprocess
    import tempfile
    import check_top_vgpr
    readelf = os.path.join(check_top_vgpr.LLVM, "llvm-readelf")
    out = {}
    for elf in check_top_vgpr.code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".o") as f:
            f.write(elf)
            f.flush()
            txt = subprocess.run([readelf, "--notes", f.name], capture_output=True, text=True).stdout
        name, args, cur = None, [], None
        for line in txt.splitlines():
            s = line.strip()
            if s.startswith("- .args:") or s.startswith(".args:"):
                args, cur = [], None
            if s.startswith("- .offset:") or s.startswith(".offset:"):
                cur = {"offset": int(s.split(":")[1])}
                args.append(cur)
            elif s.startswith(".size:") and cur is not None:
                cur["size"] = int(s.split(":")[1])
            elif s.startswith(".value_kind:") and cur is not None:
                cur["kind"] = s.split(":")[1].strip()
            elif s.startswith(".name:") and not s.startswith(".name: "):
                pass
            elif s.startswith(".symbol:"):
                out[s.split(":")[1].strip()[:-3]] = [(a["offset"], a.get("size"), a.get("kind")) for a in args]
    return out


def test_late_kernel_arguments_are_where_the_kernels_read_them():
    """alp_device.hpp: late_kernel_arg reads five arguments of the single-pass encode kernels from the kernarg segment by BYTE OFFSET (kArgDescs = 16 ..
    kArgExcCap = 64).  Nothing in the language ties those numbers to the kernels' signatures: this reads the argument layout the
    compiler recorded in the built library's code objects and checks that argument 2, 3, 4 (pointers) and 7, 8 (capacities) of k_encode_lean and
    of the single-pass k_encode_fused_f32 sit at exactly those offsets, 8 bytes each"""
    import re
    import check_top_vgpr
    lib = os.path.join(ROOT, "alp_amd", "libalpgpu.so")
    if not os.path.exists(lib) or not os.path.exists(os.path.join(check_top_vgpr.LLVM, "llvm-readelf")):
        pytest.skip("library or llvm-readelf missing")
    hdr = open(os.path.join(ROOT, "alp_amd", "csrc", "alp_device.hpp")).read()
    m = re.search(r"kArgDescs = (\d+), kArgPacked = (\d+), kArgExcs = (\d+), kArgPackedCap = (\d+), kArgExcCap = (\d+)", hdr)
    want = [int(x) for x in m.groups()]
    kernels = _kernel_args_from_library(lib)
    seen = 0
    for name, args in kernels.items():
        if "k_encode_lean" in name or ("k_encode_fused_f32ILi0" in name):
            explicit = [a for a in args if a[2] in ("global_buffer", "by_value")]
            got = [explicit[i][0] for i in (2, 3, 4, 7, 8)]
            assert got == want, (name, got, want)
            assert all(explicit[i][1] == 8 for i in (2, 3, 4, 7, 8)), name
            assert [explicit[i][2] for i in (2, 3, 4, 7, 8)] == ["global_buffer"] * 3 + ["by_value"] * 2, name
            seen += 1
    assert seen >= 3, f"k_encode_lean<false>, <true> and k_encode_fused_f32<single pass> expected, saw {seen}: {[k for k in kernels if 'encode' in k]}"


@pytest.mark.skipif(not os.path.isdir("/root/reference/include"), reason="the reference tree exists in the build container only")
def test_the_references_own_callers_compile_against_this_repos_headers():
    """the drop-in claim, pinned against header drift: the reference's unmodified test/test_alp_sample.cpp and
    publication/source_code/bench_compression_ratio/alp.cpp pass the compiler's front end with THIS repo's include/ in place of the reference's
    (gtest and the reference's data headers are outside the path: a stub gtest for syntax only, the data headers from where they lie)"""
    import subprocess
    import tempfile
    cxx = shutil.which("clang++") or "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(cxx) and shutil.which(cxx) is None:
        pytest.skip("no clang++")
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "gtest"))
        open(os.path.join(tmp, "gtest", "gtest.h"), "w").write(
            "#pragma once\n#include <iostream>\nnamespace testing { struct Test { virtual ~Test() = default; virtual void SetUp() {} virtual void TearDown() {} }; }\n"
            "struct gtest_sink_ { template <class T> gtest_sink_& operator<<(const T&) { return *this; } };\n"
            "#define TEST_F(a, b) struct a##_##b : a { void body(); }; void a##_##b::body()\n"
            "#define ASSERT_EQ(a, b) if ((a) == (b)) {} else gtest_sink_()\n#define EXPECT_EQ(a, b) if ((a) == (b)) {} else gtest_sink_()\n"
            "#define ASSERT_TRUE(a) if (a) {} else gtest_sink_()\n#define EXPECT_TRUE(a) if (a) {} else gtest_sink_()\n#define SUCCEED() gtest_sink_()\n"
            "#define ASSERT_NE(a, b) if ((a) != (b)) {} else gtest_sink_()\n#define ASSERT_LE(a, b) if ((a) <= (b)) {} else gtest_sink_()\n")
        for src, extra in (("/root/reference/test/test_alp_sample.cpp", ["-I/root/reference/data/include", "-I/root/reference/test/include"]),
                           ("/root/reference/publication/source_code/bench_compression_ratio/alp.cpp", ["-I/root/reference/data/include", "-I/root/reference/publication/source_code/include"])):
            if not os.path.exists(src):
                pytest.skip(f"{src} is not in this copy of the reference")
            cmd = [cxx, "-std=c++17", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), "-I" + tmp, *extra,
                   "-DCMAKE_SOURCE_DIR=\"/root/reference\"", "-DALP_CMAKE_SOURCE_DIR=\"/root/reference\"", "-DPAPER_ROOT=\"/root/reference/publication\"", src]
            p = subprocess.run(cmd, capture_output=True, text=True)
            assert p.returncode == 0, " ".join(cmd) + "\n" + p.stderr[-3000:]
