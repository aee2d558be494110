#!/bin/bash
# usage: tools/trace_encode.sh <tag> [env...]: rocprofv3 kernel trace of tools/time_encode.py (mixed column): start / end of every kernel of the LAST
# asynchronous alpgpu_encode_f64 call relative to its first kernel -> gpurun_out/trace_<tag>.txt
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$TAG; rm -rf $OUT; mkdir -p $OUT
( cd $GRAFT_REPO_ROOT && env "$@" rocprofv3 --kernel-trace --output-format csv -d $OUT/a -- python tools/time_encode.py 1048576 mixed > $OUT/cmd.log 2>&1 )
python - <<PY | tee $GRAFT_REPO_ROOT/gpurun_out/trace_$TAG.txt
import csv, glob
rows = []
for f in glob.glob("$OUT/**/*_kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the last launch of the persistent search (grid = about the CU count: the SECOND async init launch of a call)
idx = [i for i, r in enumerate(rows) if "k_rowgroup_init" in r["Kernel_Name"] and "ELi4ELb1E" in r["Kernel_Name"].replace(" ", "") or ("k_rowgroup_init" in r["Kernel_Name"] and ", 4, true" in r["Kernel_Name"])]
print(open("$OUT/cmd.log").read().strip().splitlines()[-1][:400])
if idx:
    last = idx[-1]
    first = last - 2 if last >= 2 else 0
    t0 = int(rows[first]["Start_Timestamp"])
    for r in rows[first:last + 14]:
        print("%9.1f %9.1f us  %-60s grid=%s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r["Kernel_Name"].replace("alpgpu::", "")[:60], r.get("Grid_Size", "?")))
else:
    print("no persistent search launch found; kernels:", sorted(set(n[:50] for n in names)))
PY
