#!/bin/bash
# usage: tools/pmc_ab.sh <kind> <n_vectors> <lib path relative to the repo root>...
# dynamic instruction counts, wave-cycles and HBM bytes of the encode kernels for A/B builds (ALPGPU_LIB), one line per kernel
cd /tmp && export TMPDIR=/tmp
KIND=$1; N=$2; shift; shift
for LIB in "$@"; do
  TAG=$(basename $LIB .so)
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcab_${TAG}_$KIND; rm -rf $OUT; mkdir -p $OUT
  ALPGPU_LIB=$GRAFT_REPO_ROOT/$LIB rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/a -- python $GRAFT_REPO_ROOT/tools/prof_encode.py $KIND $N > /dev/null 2>&1
  ALPGPU_LIB=$GRAFT_REPO_ROOT/$LIB rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/b -- python $GRAFT_REPO_ROOT/tools/prof_encode.py $KIND $N > /dev/null 2>&1
  ALPGPU_LIB=$GRAFT_REPO_ROOT/$LIB rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/c -- python $GRAFT_REPO_ROOT/tools/prof_encode.py $KIND $N > /dev/null 2>&1
  ALPGPU_LIB=$GRAFT_REPO_ROOT/$LIB rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/d -- python $GRAFT_REPO_ROOT/tools/prof_encode.py $KIND $N > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "alpgpu" in k and "finish" not in k:
            kn = k.split("(")[0].replace("void alpgpu::", "").split("<")[0]
            acc[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[kn].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for kn in acc:
    w = sum(acc[kn]["SQ_WAVES"]) / len(acc[kn]["SQ_WAVES"])
    print("$TAG $KIND", kn, "dur_us=%.0f" % sorted(dur[kn])[len(dur[kn])//2], "waves=%d" % w, " ".join("%s/wave=%.1f" % (c, sum(v)/len(v)/w) for c, v in sorted(acc[kn].items()) if c != "SQ_WAVES"))
PY
done
