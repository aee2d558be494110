#!/bin/bash
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
for exc in 0 5; do
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/a_$exc -- python $GRAFT_REPO_ROOT/tools/prof_one.py 16 $exc 1048576 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/b_$exc -- python $GRAFT_REPO_ROOT/tools/prof_one.py 16 $exc 1048576 > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections, os
out = "$OUT"
for d in sorted(glob.glob(out + "/[ab]_*")):
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list); dur = []
        for r in csv.DictReader(open(f)):
            if "k_decode_column" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        print(os.path.basename(d), "dur_us(med)=%.0f" % (sorted(dur)[len(dur)//2] if dur else -1), " ".join("%s=%.3g" % (k, sum(v)/len(v)) for k, v in sorted(acc.items())))
PY
