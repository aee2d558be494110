#!/usr/bin/env python3
"""kernel_resources.py [library] [name filter]: registers, spills, scratch and LDS of the kernels in a built libalpgpu.so, from the AMDGPU metadata of its code objects"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_top_vgpr  # noqa: E402

lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "alp_amd", "libalpgpu.so")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
readelf = os.path.join(check_top_vgpr.LLVM, "llvm-readelf")
for elf in check_top_vgpr.code_objects(lib):
    with tempfile.NamedTemporaryFile(suffix=".o") as f:
        f.write(elf)
        f.flush()
        txt = subprocess.run([readelf, "--notes", f.name], capture_output=True, text=True).stdout
    for blk in txt.split("- .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]  # noqa: E731
        name = g("name")
        if flt in name:
            print(f"{name[:110]:110s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} vspill {g('vgpr_spill_count'):>3s} sspill {g('sgpr_spill_count'):>3s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}")
