#!/bin/bash
# usage: tools/pmc_kernels.sh <tag> <command...>: dynamic instruction counts per wavefront and wave-cycles of every alpgpu kernel the
# command launches (rocprofv3 --pmc; one counter pass).  Output: gpurun_out/pmck_<tag>.txt
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmck_$TAG; rm -rf $OUT; mkdir -p $OUT
( cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_BUSY_CYCLES --output-format csv -d $OUT/a -- "$@" > $OUT/cmd.log 2>&1 )
python - <<PY | tee $GRAFT_REPO_ROOT/gpurun_out/pmck_$TAG.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "alpgpu" not in k: continue
        k = k.replace("void ", "").replace("alpgpu::", "")[:70]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "SQ_WAVES": dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in acc:
    w = sum(acc[k]["SQ_WAVES"]) / len(acc[k]["SQ_WAVES"])
    d = sorted(dur[k])[len(dur[k]) // 2]
    print("%-72s launches=%d dur_us=%.0f waves=%.0f" % (k, len(dur[k]), d, w), " ".join("%s=%.0f" % (c.replace("SQ_INSTS_", "").replace("SQ_", ""), sum(v) / len(v) / w) for c, v in sorted(acc[k].items()) if c != "SQ_WAVES"))
PY
