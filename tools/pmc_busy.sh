#!/bin/bash
# usage: tools/pmc_busy.sh <tag> <command...>: per alpgpu kernel of the command — dynamic VALU / SALU / LDS instructions per wavefront and the SQ busy /
# wait counters (two rocprofv3 --pmc passes) -> gpurun_out/pmcb_<tag>.txt
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcb_$TAG; rm -rf $OUT; mkdir -p $OUT
( cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/a -- "$@" > $OUT/a.log 2>&1 )
( cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS --output-format csv -d $OUT/b -- "$@" > $OUT/b.log 2>&1 )
python - <<PY | tee $GRAFT_REPO_ROOT/gpurun_out/pmcb_$TAG.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "alpgpu" not in k: continue
        k = k.replace("void ", "").replace("alpgpu::", "")[:44]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "SQ_WAVES": dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in acc:
    w = sum(acc[k]["SQ_WAVES"]) / len(acc[k]["SQ_WAVES"])
    d = sorted(dur[k])[len(dur[k]) // 2]
    print("%-46s dur_us=%.0f waves=%.0f" % (k, d, w), " ".join("%s=%.0f" % (c.replace("SQ_INSTS_", "").replace("SQ_", ""), sum(v) / len(v) / w) for c, v in sorted(acc[k].items()) if c != "SQ_WAVES"))
PY
