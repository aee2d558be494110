#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>      e.g. r02
# rocprofv3 --kernel-trace --stats and, in SEPARATE passes, --pmc FETCH_SIZE / --pmc WRITE_SIZE (MI355X_MICROARCH.md §HBM) for
#   dec : python bench.py --steps 20 --warmup 10 --no-extras          (k_decode_column on the benchmark column)
#   enc : python tools/prof_encode.py mixed 1048576                    (k_rowgroup_init, k_encode_fused)
#   encf: python tools/prof_encode_f32.py decimal1 1048576             (k_rowgroup_init<f32>, k_encode_fused_f32) + float decode
#   encrd: python tools/prof_encode.py rd 1048576                      (the all-ALP_RD encode)
#   cons: python tools/prof_consumers.py 0 1048576                     (k_sink_direct, k_decode_column<2, false, kSinkSum>, k_consume_column: the fused SUM consumer, all three kernels)
#   narrow: python tools/time_one.py 8:1048576:2                         (k_decode_column<2, true>: the two-vectors-per-workgroup decode of a narrow column)
#   sinkf: python tools/prof_sink_direct_f32.py 1048576                  (k_sink_direct_f32 and the staged float SUM sink)
#   shapes: python tools/prof_decode_shapes.py 1048576                   (the auto rule's other decode launches: k_decode_pairs, two vectors x six workgroups per CU, one x six)
#   ahead: python tools/prof_read_ahead.py 1048576                       (k_read_ahead beside k_decode_column<1> / <2> on 3- and 4-bit columns; kernel stats only mean something without --pmc)
#   streamf: BWS=4 EXCS=0 SHAPES=2 SIZES=1048576 python tools/time_f32_narrow.py   (a 4-bit float column: k_decode_column_f32<2> and, by the rule, k_decode_stream_f32<12, 24576, 2, 12>)
# (enc / encrd also decode what they encoded: the ALP_RD column's k_decode_column row; encf runs the float search in front of the float encode)
# raw outputs under gpurun_out/<tag>_prof/, condensed by tools/summarize_round.py into profiles/<tag>_* (run the summary LOCALLY on the
# merged directory, and remove a stale local gpurun_out/<tag>_prof first: rocprofv3 names its files after process ids, a second run
# does not overwrite the first)
# tools/profile_round.sh <tag> <name>... re-runs only the named commands (into the same directory; fetch the directory back and summarize again)
TAG=$1; shift
ONLY="$*"
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${TAG}_prof
if [ -z "$ONLY" ]; then rm -rf $OUT; fi
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, command...
  name=$1; shift
  if [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $name "; then return; fi
  rm -rf $OUT/${name}_stats $OUT/${name}_fetch $OUT/${name}_write
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${name}_stats -- "$@" > $OUT/${name}_stats.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${name}_fetch -- "$@" > $OUT/${name}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${name}_write -- "$@" > $OUT/${name}_write.log 2>&1
}
run dec python $ROOT/bench.py --steps 20 --warmup 10 --no-extras
run enc python $ROOT/tools/prof_encode.py mixed 1048576
run encf python $ROOT/tools/prof_float.py 1048576
run encrd python $ROOT/tools/prof_encode.py rd 1048576
run cons python $ROOT/tools/prof_consumers.py 0 1048576
run narrow python $ROOT/tools/time_one.py 8:1048576:2
run sinkf python $ROOT/tools/prof_sink_direct_f32.py 1048576
run shapes python $ROOT/tools/prof_decode_shapes.py 1048576
run ahead python $ROOT/tools/prof_read_ahead.py 1048576
BWS=4 EXCS=0 SHAPES=2 SIZES=1048576 run streamf python $ROOT/tools/time_f32_narrow.py
cd $ROOT
if [ -f $OUT/dec_stats.log ]; then grep -h '"metric"' $OUT/dec_stats.log | tail -1 > $OUT/bench_under_rocprof.json; fi
python tools/summarize_round.py $TAG $OUT
