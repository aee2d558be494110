#!/bin/bash
# usage: tools/pmc_encode.sh <outdir> <kind>; SQ counters for the encode kernels
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/a -- python $GRAFT_REPO_ROOT/tools/prof_encode.py $2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python $GRAFT_REPO_ROOT/tools/prof_encode.py $2 > /dev/null 2>&1
python - <<PY
import csv, glob, collections, os
out = "$OUT"
for d in sorted(glob.glob(out + "/[ab]")):
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "alpgpu" in k:
                kn = k.split("(")[0].replace("void alpgpu::", "")
                acc[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur[kn].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for kn in acc:
            print(kn, "dur_us=%.0f" % sorted(dur[kn])[len(dur[kn])//2], " ".join("%s=%.3g" % (c, sum(v)/len(v)) for c, v in sorted(acc[kn].items())))
PY
