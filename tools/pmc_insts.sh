#!/bin/bash
# usage: tools/pmc_insts.sh <kind> <n_vectors> <lib>...   dynamic instruction counts per wavefront of the encode kernels for A/B builds
cd /tmp && export TMPDIR=/tmp
KIND=$1; N=$2; shift; shift
for LIB in "$@"; do
  TAG=$(basename $LIB .so)
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmci_${TAG}_$KIND; rm -rf $OUT; mkdir -p $OUT
  ALPGPU_LIB=$GRAFT_REPO_ROOT/$LIB rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $OUT/a -- python $GRAFT_REPO_ROOT/tools/prof_encode.py $KIND $N > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_encode_fused" in k:
            kn = "fused"
            acc[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[kn].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for kn in acc:
    w = sum(acc[kn]["SQ_WAVES"]) / len(acc[kn]["SQ_WAVES"])
    print("%-10s $KIND" % "$TAG", "dur_us=%.0f" % sorted(dur[kn])[len(dur[kn])//2], " ".join("%s=%.0f" % (c.replace("SQ_INSTS_", "").replace("SQ_", ""), sum(v)/len(v)/w) for c, v in sorted(acc[kn].items()) if c != "SQ_WAVES"))
PY
done
