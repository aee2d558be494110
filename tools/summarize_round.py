#!/usr/bin/env python3
"""Condense tools/profile_round.sh outputs (gpurun_out/<tag>_prof) into small tracked files under profiles/:
  <tag>_kernel_stats.csv  per-kernel rocprofv3 --stats rows of the three commands (alpgpu kernels only)
  <tag>_hbm_traffic.json  FETCH_SIZE / WRITE_SIZE per launch of every alpgpu kernel, with the gfx950 corrections
  hbm_traffic.json        the decode kernel's figure + the sha-256 prefix of the library it was measured with (bench.py quotes
                          it as roofline.traffic only when that matches the library it runs)
usage: tools/summarize_round.py <tag> <dir>"""
import collections, csv, glob, hashlib, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, d = sys.argv[1], sys.argv[2]
prof = os.path.join(ROOT, "profiles")
sha = hashlib.sha256(open(os.path.join(ROOT, "alp_amd", "libalpgpu.so"), "rb").read()).hexdigest()[:16]
try:
    head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    head = "(no .git on the GPU box: see the commit that adds this file)"


def short(name):
    name = name.replace("void ", "")
    return name if len(name) < 110 else name[:100] + "..."


cmds = {"dec": "python bench.py --steps 20 --warmup 10 --no-extras", "enc": "python tools/prof_encode.py mixed 1048576", "encf": "python tools/prof_float.py 1048576",
        "encrd": "python tools/prof_encode.py rd 1048576", "cons": "python tools/prof_consumers.py 0 1048576", "narrow": "python tools/time_one.py 8:1048576:2", "sinkf": "python tools/prof_sink_direct_f32.py 1048576", "shapes": "python tools/prof_decode_shapes.py 1048576", "ahead": "python tools/prof_read_ahead.py 1048576", "streamf": "BWS=4 EXCS=0 SHAPES=2 SIZES=1048576 python tools/time_f32_narrow.py"}
with open(os.path.join(prof, f"{tag}_kernel_stats.csv"), "w") as w:
    w.write(f"# rocprofv3 --kernel-trace --stats --output-format csv; library sha256[:16] = {sha}; tree = {head}\n")
    w.write("Command,Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev\n")
    for key, cmd in cmds.items():
        fs = glob.glob(os.path.join(d, f"{key}_stats", "**", "*_kernel_stats.csv"), recursive=True)
        if not fs:
            continue
        for r in csv.DictReader(open(fs[0])):
            if "alpgpu::" in r["Name"]:
                w.write(",".join([json.dumps(cmd), json.dumps(short(r["Name"]))] + [r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")]) + "\n")
print(open(os.path.join(prof, f"{tag}_kernel_stats.csv")).read())

traffic = {"library_sha256_16": sha, "tree": head,
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (KB per launch, mean over launches); "
                     "hbm_bytes = 2 x FETCH_SIZE (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md §HBM) + WRITE_SIZE", "kernels": {}}
for key, cmd in cmds.items():
    res = {}
    for ctr, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        fs = glob.glob(os.path.join(d, f"{key}_{sub}", "**", "*_counter_collection.csv"), recursive=True)
        if not fs:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            if r["Counter_Name"] == ctr and "alpgpu::" in r["Kernel_Name"]:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            res.setdefault(k, {})[ctr + "_KB_mean"] = sum(v) / len(v)
            res[k]["launches"] = len(v)
    for k, v in res.items():
        if v.get("FETCH_SIZE_KB_mean", 0) + v.get("WRITE_SIZE_KB_mean", 0) < 64:  # the gated recovery kernels and the like
            continue
        v["hbm_bytes_per_launch"] = int(2 * 1024 * v.get("FETCH_SIZE_KB_mean", 0) + 1024 * v.get("WRITE_SIZE_KB_mean", 0))
        traffic["kernels"].setdefault(cmd, {})[k] = v
json.dump(traffic, open(os.path.join(prof, f"{tag}_hbm_traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1)[:3000])
dec = [v for k, v in traffic["kernels"].get(cmds["dec"], {}).items() if "k_decode_column" in k]
if dec:
    json.dump({"kernel": "k_decode_column", "vectors": 1 << 20, "hbm_bytes_per_launch": dec[0]["hbm_bytes_per_launch"], "lib_sha16": sha,
               "source": f"profiles/{tag}_hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `{cmds['dec']}`; "
                         "FETCH_SIZE x2 (gfx950 wide-read correction, MI355X_MICROARCH.md), WRITE_SIZE x1"},
              open(os.path.join(prof, "hbm_traffic.json"), "w"), indent=1)
b = os.path.join(d, "bench_under_rocprof.json")
if os.path.exists(b) and os.path.getsize(b):
    open(os.path.join(prof, f"{tag}_bench_under_rocprof.json"), "w").write(open(b).read())
