#!/bin/bash
# usage: tools/pmc_variants.sh <kind> <lib.so>...   dynamic instruction counts of k_encode_fused for A/B builds (ALPGPU_LIB)
cd /tmp && export TMPDIR=/tmp
KIND=$1; shift
for LIB in "$@"; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcv_$(basename $LIB .so); rm -rf $OUT; mkdir -p $OUT
  ALPGPU_LIB=$GRAFT_REPO_ROOT/alp_amd/$LIB rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/tools/prof_encode.py $KIND 262144 > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_encode_fused" in k:
            acc["fused"][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur["fused"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for kn in acc:
    w = sum(acc[kn]["SQ_WAVES"]) / len(acc[kn]["SQ_WAVES"])
    print("$LIB", "dur_us=%.0f" % sorted(dur[kn])[len(dur[kn])//2], " ".join("%s/wave=%.1f" % (c, sum(v)/len(v)/w) for c, v in sorted(acc[kn].items()) if c != "SQ_WAVES"))
PY
done
