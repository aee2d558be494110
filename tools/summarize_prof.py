#!/usr/bin/env python3
"""Condense rocprofv3 outputs under gpurun_out/ into small tracked files under profiles/.
usage: tools/summarize_prof.py <tag> <stats_dir> [<pmc_fetch_dir> <pmc_write_dir> <vectors>]"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("void ", "")
    return name if len(name) < 100 else name[:60] + "...<trimmed>"


def main():
    tag, stats_dir = sys.argv[1], sys.argv[2]
    out = os.path.join(ROOT, "profiles")
    f = glob.glob(os.path.join(stats_dir, "**", "*_kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w") as w:
        w.write("# rocprofv3 --kernel-trace --stats --output-format csv -- " + os.environ.get("PROF_CMD", "python bench.py --steps 10 --warmup 2 --no-extras") + "\n")
        w.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev\n")
        for r in rows:
            w.write(",".join([json.dumps(short(r["Name"]))] + [r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")]) + "\n")
    print(open(os.path.join(out, f"{tag}_kernel_stats.csv")).read())
    if len(sys.argv) > 5:
        fd, wd, vectors = sys.argv[3], sys.argv[4], int(sys.argv[5])
        res = {}
        for key, d in (("FETCH_SIZE", fd), ("WRITE_SIZE", wd)):
            f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == key:
                    acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
            res[key] = {k: {"launches": len(v), "mean_KB": sum(v) / len(v)} for k, v in acc.items() if "alpgpu" in k}
        kern = [k for k in res["FETCH_SIZE"] if "k_decode_column" in k][0]
        fetch_kb, write_kb = res["FETCH_SIZE"][kern]["mean_KB"], res["WRITE_SIZE"][kern]["mean_KB"]
        # MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced
        # streaming read -> double it; WRITE_SIZE here equals the known output size exactly (n_vectors * 8192 B), factor 1.
        traffic = fetch_kb * 1024 * 2 + write_kb * 1024
        summary = {"kernel": "k_decode_column", "vectors": vectors, "hbm_bytes_per_launch": int(traffic),
                   "fetch_size_KB_raw": fetch_kb, "write_size_KB_raw": write_kb,
                   "source": f"profiles/{tag}_pmc.json: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; "
                             "FETCH_SIZE x2 (gfx950 wide-read correction, MI355X_MICROARCH.md), WRITE_SIZE x1 (matches the known 8192 B/vector)",
                   "all_alpgpu_kernels": res}
        json.dump(summary, open(os.path.join(out, f"{tag}_pmc.json"), "w"), indent=1)
        json.dump({k: summary[k] for k in ("kernel", "vectors", "hbm_bytes_per_launch", "source")}, open(os.path.join(out, "hbm_traffic.json"), "w"), indent=1)
        print(json.dumps(summary, indent=1)[:1500])


if __name__ == "__main__":
    main()
