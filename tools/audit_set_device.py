#!/usr/bin/env python3
"""audit_set_device.py: every entry point of the C ABI that is handed a context must make that context's device current ON THE CALLING THREAD
before it touches the HIP runtime (hipSetDevice is per host thread: a process that drives eight GPUs from eight threads, or from one thread
in turn, otherwise launches on whatever device the thread used last).  Parses alp_amd/csrc/api_*.hip (the translation units behind include/alpgpu.h): a function "sets the device" if its body
has ALPGPU_CHECK_CTX / hipSetDevice, or if the FIRST thing it does with its context is to hand it to a function that does.  Lists every
exported alpgpu_* function with a context parameter and how it is covered; exit status 1 if one is not.  (tests/test_build_rules.py runs it.)"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# entry points that by design do not touch the device (documented in include/alpgpu.h) or only read host-side fields of the context
HOST_ONLY = {"alpgpu_set_stream", "alpgpu_use_own_stream", "alpgpu_set_option", "alpgpu_decode_vectors_per_wg", "alpgpu_decode_reads_ahead", "alpgpu_decode_runs", "alpgpu_decode_runs_f32", "alpgpu_device_info"}


def functions(text):
    """name -> (signature, body) of every function definition at file / extern-C / namespace scope (brace matching)"""
    out = {}
    for m in re.finditer(r"^(?:static\s+|template\s*<[^>]*>\s*)*(?:[\w:<>\*&]+\s+)+\**(\w+)\s*\(([^;{}]*?)\)\s*\{", text, re.M):
        name, args = m.group(1), m.group(2)
        if name in ("if", "for", "while", "switch", "return", "sizeof"):
            continue
        depth, i = 1, m.end()
        while depth and i < len(text):
            depth += {"{": 1, "}": -1}.get(text[i], 0)
            i += 1
        out.setdefault(name, (args, text[m.end():i]))
    return out


def main():
    csrc = os.path.join(ROOT, "alp_amd", "csrc")
    text = "\n".join(open(os.path.join(csrc, f)).read() for f in sorted(os.listdir(csrc)) if f.startswith("api_") and f.endswith(".hip"))
    text = re.sub(r"//[^\n]*", "", text)
    fns = functions(text)
    direct = {n for n, (_, b) in fns.items() if "ALPGPU_CHECK_CTX(" in b or "hipSetDevice(" in b or "ALPGPU_PRIM(" in b}  # (ALPGPU_PRIM starts with ALPGPU_CHECK_CTX)
    covered = dict((n, "sets the device itself") for n in direct)
    changed = True
    while changed:  # a function is covered if it calls a covered function before any other use of the HIP runtime
        changed = False
        for n, (_, b) in fns.items():
            if n in covered:
                continue
            calls = [(m.start(), m.group(1)) for m in re.finditer(r"\b(\w+)\s*(?:<[^;(){}]*>)?\s*\(", b)]
            first_hip = min([p for p, c in calls if c.startswith("hip") and c not in ("hipGetLastError", "hipGetErrorString")] + [len(b)])
            via = [c for p, c in calls if c in covered and c != n and p < first_hip]
            if via:
                covered[n] = "through " + via[0]
                changed = True
    bad = 0
    exported = [n for n, (a, _) in fns.items() if n.startswith("alpgpu_") and re.search(r"alpgpu_ctx\s*\*", a)]
    for n in sorted(exported):
        how = covered.get(n) or ("host-side only" if n in HOST_ONLY else None)
        print(f"{n:44s} {how or 'NOT COVERED'}")
        bad += how is None
    print(f"{len(exported)} entry points take a context; {bad} of them may run on the wrong device")
    if len(exported) < 60:
        print("fewer entry points than the ABI has: the parser missed definitions")
        return 2
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
