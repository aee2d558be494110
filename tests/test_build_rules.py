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
    in alp_amd/csrc/api.hip (the in-process multi-GPU path has never met a second physical GPU)"""
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
