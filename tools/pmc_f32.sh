#!/bin/bash
# usage: tools/pmc_f32.sh <kind>...   dynamic instruction counts per wavefront of k_encode_fused_f32
cd /tmp && export TMPDIR=/tmp
for KIND in "$@"; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcf_$KIND; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/tools/prof_encode_f32.py $KIND 262144 2>/dev/null | tail -1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for tag in ("k_encode_fused_f32", "k_rowgroup_init"):
            if tag in k:
                acc[tag][r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur[tag].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for kn in acc:
    w = sum(acc[kn]["SQ_WAVES"]) / len(acc[kn]["SQ_WAVES"])
    print("$KIND", kn, "dur_us=%.0f" % sorted(dur[kn])[len(dur[kn])//2], " ".join("%s/wave=%.1f" % (c, sum(v)/len(v)/w) for c, v in sorted(acc[kn].items()) if c != "SQ_WAVES"))
PY
done
