#!/usr/bin/env python3
"""check_top_vgpr.py [hipcc flags...]: compiles every kernel source to assembly and lists the kernels in which a 64-bit instruction (v_*_b64 /
_f64 / _i64 / _u64) reads the LAST register of the kernel's allocation as a single-register operand — e.g. the shift amount of v_lshrrev_b64.
On gfx950 such an operand is range-checked as if it were a register pair: v(N-1):vN falls out of the allocation and the instruction reads
VGPR0 instead (tools/last_vgpr_probe.hip shows it in isolation; profiles/r03_consumers.txt tells how it was found).  Exit status 1 if any
kernel has the pattern.  The register allocator does not know this rule; a kernel that uses exactly 8k registers can get it from any rebuild."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-I" + os.path.join(ROOT, "include")]


def kernels_with_pattern(asm_text):
    out = []
    for part in re.split(r"\n(?=_Z[\w]+:\s)", asm_text):
        m = re.match(r"(_Z\w+):", part)
        nv = re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", part)
        if not m or not nv:
            continue
        n = int(nv.group(1))
        if n % 8 != 0:  # the allocation is rounded up to 8: the last allocated register is not one the kernel uses
            continue
        top = "v%d" % (n - 1)
        hits = []
        for line in part.split("\n"):
            t = line.split(";")[0].strip()
            if not t.startswith("v_"):
                continue
            op = t.split()[0]
            if not re.search(r"(_b64|_f64|_i64|_u64)", op):
                continue
            operands = [o.strip() for o in t[len(op):].split(",")]
            if top in operands[1:]:
                hits.append(t)
        if hits:
            out.append((m.group(1), n, hits))
    return out


def main(extra_flags=()):
    bad = kernels = 0
    for src in sorted(glob.glob(os.path.join(ROOT, "alp_amd", "csrc", "*.hip"))):
        run = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra_flags, "-S", "--cuda-device-only", "-o", "-", src], capture_output=True, text=True)
        if run.returncode != 0:
            print(f"{os.path.basename(src)}: hipcc failed: {run.stderr[-400:]}")
            return 2
        asm = run.stdout
        kernels += len(re.findall(r"\.amdhsa_next_free_vgpr", asm))
        for name, n, hits in kernels_with_pattern(asm):
            bad += 1
            print(f"{os.path.basename(src)}: {name} ({n} registers): {len(hits)} instruction(s), e.g. {hits[0]}")
    print(f"{bad} of {kernels} kernels read the last register of their allocation in a 64-bit instruction")
    if kernels < 40:
        print("fewer kernels than the library has: the check did not see the real build")
        return 2
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
