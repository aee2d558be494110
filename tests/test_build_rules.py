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


def test_committed_profiles_are_those_of_the_built_library():
    """profiles/hbm_traffic.json (what bench.py quotes as roofline.traffic) and profiles/r03_hbm_traffic.json carry the sha-256 prefix of the
    library they were measured with; the library in the tree is that one (builds are deterministic: same sources, same compiler, same bytes)"""
    import hashlib
    import json
    lib = os.path.join(ROOT, "alp_amd", "libalpgpu.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    sha = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]
    assert json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))["lib_sha16"] == sha
    assert json.load(open(os.path.join(ROOT, "profiles", "r03_hbm_traffic.json")))["library_sha256_16"] == sha
