#!/bin/bash
# r06_call.sh <step>: ONE gpurun call of round 6 (every step under its own wall-clock guard; logs under gpurun_out/r06/<step>/).
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/r06_call.sh 1'
step=$1
out=gpurun_out/r06/$step
mkdir -p $out
run() { # seconds, log, command...
  t=$1; log=$2; shift 2
  echo "== $* (limit ${t}s)" | tee -a $out/$log
  timeout $t "$@" >> $out/$log 2>&1
  echo "== rc $?" | tee -a $out/$log
}
case $step in
1) # first look: the whole GPU suite on the split library + round 6's decode policy; the float sweep; what an unhinted decode costs; the self-stretching lead
  run 600 tests.txt python -m pytest tests -m gpu -x -q
  run 400 f32_sweep.txt python tools/sweep_f32_decode.py
  run 200 unhinted.txt python tools/time_unhinted.py
  ALPGPU_READ_AHEAD_ADAPT=0 run 120 adapt_off.txt python tools/time_read_ahead.py
  ALPGPU_READ_AHEAD_ADAPT=1 run 120 adapt_on.txt python tools/time_read_ahead.py
  run 300 bench.txt python bench.py
  ;;
2) # the whole GPU suite again; the tiles' search items (experiment: vectors alone with / without them); the decode of encoder output in a grid of shapes
  run 600 tests.txt python -m pytest tests -m gpu -x -q
  run 300 encode.txt python tools/time_encode.py
  run 300 decode_encoded.txt python tools/time_decode_encoded.py
  run 200 unhinted.txt python tools/time_unhinted.py
  ;;
3) # the tiles' search items split over wavefronts (second form); the float store decode with one wavefront per vector; the decode grid again (the empty-vectors rule)
  run 300 encode.txt python tools/time_encode.py
  WIDTHS=1,2,3,4,5,6,7,8,10,12,16,20,24,28,32 run 400 f32_sweep.txt python tools/sweep_f32_decode.py
  PADS=0,6,14 run 200 decode_encoded.txt python tools/time_decode_encoded.py
  run 300 tests.txt python -m pytest tests/test_float_gpu.py tests/test_unhinted_gpu.py tests/test_decode_gpu.py -m gpu -x -q
  ;;
4) # the persistent, software-pipelined encode: parity under both instances, then its timing; the float / unhinted tests that failed in call 3
  ALPGPU_ENCODE_PIPELINED=3 run 400 tests_pipe3.txt python -m pytest tests/test_encode_gpu.py tests/test_async_init_gpu.py tests/test_recovery_gpu.py tests/test_reference_gpu.py -m gpu -x -q
  ALPGPU_ENCODE_PIPELINED=2 run 400 tests_pipe2.txt python -m pytest tests/test_encode_gpu.py tests/test_async_init_gpu.py tests/test_recovery_gpu.py tests/test_reference_gpu.py -m gpu -x -q
  run 300 encode.txt python tools/time_encode.py
  run 300 tests.txt python -m pytest tests/test_unhinted_gpu.py tests/test_float_gpu.py tests/test_decode_gpu.py -m gpu -x -q
  ;;
5) # the pipelined encode with tiles taken one ahead
  ALPGPU_ENCODE_PIPELINED=2 run 400 tests_pipe2.txt python -m pytest tests/test_encode_gpu.py tests/test_async_init_gpu.py tests/test_recovery_gpu.py tests/test_reference_gpu.py -m gpu -x -q
  run 300 encode.txt python tools/time_encode.py 1048576 mixed
  ALPGPU_ENCODE_PIPE_ROOMY=1 run 300 encode_roomy.txt python tools/time_encode.py 1048576 mixed
  ;;
6) # a 256-entry exception stage in the double decode (A/B library), the decode of encoder output in both; then the whole suite and the bench line on the tree as it is
  PADS=0,6 run 200 decode_encoded.txt python tools/time_decode_encoded.py
  ALPGPU_LIB=build/variants/libalpgpu_r06_exc256.so PADS=0,6 run 200 decode_encoded_exc256.txt python tools/time_decode_encoded.py
  ALPGPU_LIB=build/variants/libalpgpu_r06_exc256.so run 300 tests_exc256.txt python -m pytest tests/test_decode_gpu.py tests/test_decode_sum_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q
  run 600 tests.txt python -m pytest tests -m gpu -x -q
  run 400 bench.txt python bench.py
  ;;
7) # the 256-entry exception stage as an INSTANCE of its own, launched for exception-heavy columns only: parity, then the decode grid
  run 300 tests.txt python -m pytest tests/test_decode_gpu.py tests/test_decode_sum_gpu.py tests/test_fuzz_gpu.py tests/test_unhinted_gpu.py -m gpu -x -q
  PADS=0,6 run 200 decode_encoded.txt python tools/time_decode_encoded.py
  ;;
8) # the float store decode on narrow columns, taken apart: as built / without the unpack / a fill behind the descriptors / a plain fill
  run 200 narrow.txt python tools/time_f32_narrow.py
  for k in 1 2 3; do ALPGPU_LIB=build/variants/libalpgpu_f32_dissect$k.so EXCS=0 SHAPES=2,4 run 100 narrow_dissect$k.txt python tools/time_f32_narrow.py; done
  ;;
9) # the streaming float decode: parity, then against the small-workgroup shapes on narrow columns
  run 300 tests.txt python -m pytest tests/test_float_gpu.py -m gpu -x -q -k "streamed or shapes"
  SHAPES=2,16,17 run 300 narrow.txt python tools/time_f32_narrow.py
  ;;
10) # the streaming float decode by workgroups per CU
  for g in 1 2 3 4 5 8; do ALPGPU_STREAM_WGS_PER_CU=$g BWS=1,3,6 SHAPES=16,17 run 100 narrow_g$g.txt python tools/time_f32_narrow.py; done
  ;;
11) # the streaming float decode with a loading wavefront: parity, timing by residency
  run 300 tests.txt python -m pytest tests/test_float_gpu.py -m gpu -x -q -k "streamed or shapes"
  SHAPES=2,16,17,18 run 300 narrow.txt python tools/time_f32_narrow.py
  for g in 1 2; do ALPGPU_STREAM_WGS_PER_CU=$g BWS=3 SHAPES=16,17 run 100 narrow_g$g.txt python tools/time_f32_narrow.py; done
  for w in 6 8; do ALPGPU_LIB=build/variants/libalpgpu_stream_w$w.so BWS=3,6 SHAPES=16,17 run 100 narrow_w$w.txt python tools/time_f32_narrow.py; done
  ;;
12) # the streaming float decode with D loading wavefronts: parity, timing by depth (19: 1, 20: 2, 16: 3 loaders; 17: chunks of 16, 2 loaders) and residency
  run 300 tests.txt python -m pytest tests/test_float_gpu.py -m gpu -x -q -k "streamed or shapes"
  for g in 1 2 3; do ALPGPU_STREAM_WGS_PER_CU=$g BWS=1,3,6 SHAPES=19,20,16,17 run 100 narrow_g$g.txt python tools/time_f32_narrow.py; done
  ;;
13) # reads on the beat of the wall clock: the read-ahead (small-workgroup decode) and the streaming decode's loaders
  for b in 0 5 10 20 40; do ALPGPU_READ_AHEAD_BURST_US=$b BWS=3,6 SHAPES=2 run 100 ahead_burst$b.txt python tools/time_f32_narrow.py; done
  for b in 0 10 20 40 80; do ALPGPU_STREAM_BURST_US10=$b BWS=3,6 SHAPES=16,20 run 100 stream_burst$b.txt python tools/time_f32_narrow.py; done
  ;;
14) # the streaming decode with the read-ahead beside it
  SIZES=1048576 LEADS=10,20,40 BWS=1,3,6 SHAPES=2,16,20 run 200 stream_ahead.txt python tools/time_f32_narrow.py
  ;;
16) # the streaming decode taken apart: no record loads / no stores
  for k in 1 2; do ALPGPU_LIB=build/variants/libalpgpu_stream_dissect$k.so EXCS=0 BWS=1,3,6 SHAPES=16,20 run 100 stream_dissect$k.txt python tools/time_f32_narrow.py; done
  for g in 1 2; do ALPGPU_STREAM_WGS_PER_CU=$g ALPGPU_LIB=build/variants/libalpgpu_stream_dissect2.so EXCS=0 BWS=3 SHAPES=16,20,19 run 100 stream_dissect2_g$g.txt python tools/time_f32_narrow.py; done
  ;;
17) # the streaming float decode against the rule's choice, by width
  SIZES=1048576 BWS=1,2,3,4,5,6,7,8,9,10,11,12,14,16,18,20,24,28,32 SHAPES=16,20,18 run 400 widths.txt python tools/time_f32_narrow.py
  ;;
18) # more decoding wavefronts per workgroup, one workgroup per CU (21: 8 decoders, chunks of 16; 22: 8, chunks of 8; 23: 16 decoders, chunks of 16)
  run 300 tests.txt python -m pytest tests/test_float_gpu.py -m gpu -x -q -k "streamed"
  BWS=1,3,6,8 SHAPES=20,21,22,23 run 200 narrow.txt python tools/time_f32_narrow.py
  ALPGPU_STREAM_WGS_PER_CU=2 BWS=3,6 SHAPES=21,22,23 run 200 narrow_g2.txt python tools/time_f32_narrow.py
  ;;
19) # tables built by the decoding wavefronts; 12 / 14 decoders per workgroup
  run 300 tests.txt python -m pytest tests/test_float_gpu.py -m gpu -x -q -k "streamed"
  BWS=1,3,6,8 SHAPES=20,23,24,25,26 run 200 narrow.txt python tools/time_f32_narrow.py
  ;;
20) # 12 decoders over larger arenas: widths 2 .. 32
  run 300 tests.txt python -m pytest tests/test_float_gpu.py -m gpu -x -q -k "streamed"
  SIZES=1048576 BWS=2,4,7,8,9,10,12,14,16,20,24,28,32 SHAPES=24,27,28 run 300 widths.txt python tools/time_f32_narrow.py
  ;;
21) # how far ahead: 1 / 2 / 4 loading wavefronts (30 / 24 / 29), chunks of 12, 12 decoders
  SIZES=1048576 BWS=3,6,8 SHAPES=30,24,29 run 300 depth.txt python tools/time_f32_narrow.py
  ;;
22) # the exception span in 16-byte units
  run 300 tests.txt python -m pytest tests/test_float_gpu.py -m gpu -x -q -k "streamed"
  BWS=3,6 SHAPES=24,27 run 300 narrow.txt python tools/time_f32_narrow.py
  ;;
23) # the rule streams narrow float columns without exceptions: the whole GPU suite, the float sweep rows of the bench
  run 900 tests.txt python -m pytest tests -m gpu -x -q
  SIZES=1048576 BWS=1,2,3,4,5,6,7,8,9 SHAPES=2 run 200 rule.txt python tools/time_f32_narrow.py
  ;;
24) # the ALP_RD dictionary looked up in LDS (float kernels): parity, then SUM / decode of the bench's float columns against the lookup in registers
  run 600 tests.txt python -m pytest tests/test_float_gpu.py tests/test_decode_sum_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q
  run 200 sum.txt python tools/time_f32_sum.py
  ALPGPU_LIB=build/variants/libalpgpu_rd_dict_regs.so run 200 sum_regs.txt python tools/time_f32_sum.py
  ;;
25) # the float sinks take an ALP vector's exception values behind the quads
  run 600 tests.txt python -m pytest tests/test_float_gpu.py tests/test_decode_sum_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q
  run 200 sum.txt python tools/time_f32_sum.py
  ;;
26) # the float sink's LDS stage for vectors of any exception count (it was: at most 48)
  run 200 sum.txt python tools/time_f32_sum.py
  ALPGPU_LIB=build/variants/libalpgpu_sinkf_stage_128.so run 200 sum_128.txt python tools/time_f32_sum.py
  ALPGPU_LIB=build/variants/libalpgpu_sinkf_stage_all.so run 200 sum_all.txt python tools/time_f32_sum.py
  ;;
27) # dynamic instruction counts of the float sinks on the bench's float columns
  bash tools/pmc_busy.sh r06_sumf python tools/time_f32_sum.py 262144 > $out/pmc.txt 2>&1
  ;;
28) # the ALP_RD dictionary looked up in LDS, double kernels: parity, then decode / SUM of an ALP_RD column against the lookup in registers
  run 600 tests.txt python -m pytest tests/test_decode_gpu.py tests/test_decode_sum_gpu.py tests/test_fuzz_gpu.py tests/test_reference_gpu.py -m gpu -x -q
  run 200 rd.txt python tools/time_rd_f64.py
  ALPGPU_LIB=build/variants/libalpgpu_rd64_regs.so run 200 rd_regs.txt python tools/time_rd_f64.py
  ;;
29) # the whole GPU suite and the bench line on the tree as it is
  run 900 tests.txt python -m pytest tests -m gpu -x -q
  run 600 bench.txt python bench.py
  ;;
32) # narrow double columns under the read-ahead: two vectors per workgroup, five workgroups per CU — the suite, the rule against the old shape
  run 900 tests.txt python -m pytest tests -m gpu -x -q
  EXCS=0 BWS=1,2,3,4,5,6,7,8,9 PADS=0,11 run 400 pad.txt python tools/sweep_pad_narrow.py
  ;;
34) # the double sinks take an ALP vector's exception values behind the pairs: parity, then the SUM with 0 / 20 / 100 exceptions per vector
  run 900 tests.txt python -m pytest tests/test_decode_sum_gpu.py tests/test_fuzz_gpu.py tests/test_last_register_gpu.py tests/test_encode_gpu.py -m gpu -x -q
  run 300 sum_exc.txt python tools/time_sum_exc.py
  ;;
40) # the store decode with each XCD taking G consecutive workgroups' vectors (G = 4 / 32 / 512) against the dispatcher's round-robin, arms alternating twice
  for rep in 1 2; do
    run 200 xcd.txt python tools/time_decode_lib.py
    for g in 4 32 512; do ALPGPU_LIB=build/variants/libalpgpu_xcd$g.so run 200 xcd.txt python tools/time_decode_lib.py; done
  done
  EXC=20 run 200 xcd_exc.txt python tools/time_decode_lib.py
  EXC=20 ALPGPU_LIB=build/variants/libalpgpu_xcd32.so run 200 xcd_exc.txt python tools/time_decode_lib.py
  ;;
41) # soak: the whole GPU suite three times, smoke, the default bench line
  for i in 1 2 3; do run 600 tests_$i.txt python -m pytest tests -m gpu -x -q -p no:cacheprovider; done
  run 300 smoke.txt python -c "import __graft_entry__ as g; g.smoke()"
  timeout 600 python bench.py > $out/bench.json 2> $out/bench.err
  ;;
42) # the whole-column entry points captured into a hipGraph and replayed
  run 800 graph.txt python -m pytest tests/test_graph_gpu.py -m gpu -q -s -p no:cacheprovider
  ;;
43) # host threads: the per-vector drop-in API from six threads, the column ABI from five threads with a context each
  run 800 threads.txt python -m pytest tests/test_threads_gpu.py tests/test_dropin_gpu.py -m gpu -q -s -p no:cacheprovider -k thread
  ;;
44) # the N = 8 bench path end to end with eight ranks on ONE GPU (gloo; bookkeeping, not a measurement) -> profiles/r06_bench_8ranks_one_gpu.json
  ALPGPU_BENCH_TEST_SHARED_GPU=1 timeout 800 python bench.py --gpus 8 --column-gb 4 --steps 3 --warmup 1 > $out/bench8.json 2> $out/bench8.err
  ;;
45) # parity of two shipped options that had no test: the classic single-pass encode kernel, the decode's residency pad
  run 800 opts.txt python -m pytest tests/test_encode_gpu.py tests/test_decode_gpu.py -m gpu -q -p no:cacheprovider -k "classic or residency"
  ;;
47) # the final tree: the whole GPU suite twice, smoke
  for i in 1 2; do run 700 tests_$i.txt python -m pytest tests -m gpu -x -q -p no:cacheprovider; done
  run 300 smoke.txt python -c "import __graft_entry__ as g; g.smoke()"
  ;;
49) # fuzz campaign with fresh seeds: 20 000 double + 20 000 float columns against the oracle, 10 000 against the real reference (call 48: 2 000)
  ALPGPU_FUZZ_ROUNDS=20000 ALPGPU_FUZZ_SEED_BASE=2000000 run 2800 fuzz.txt python -m pytest tests/test_fuzz_gpu.py -m gpu -q -p no:cacheprovider -n 8
  ;;
50) # random combinations of the options over random columns, 500 s
  FUZZ_SECONDS=500 run 1200 fuzz_options.txt python tools/fuzz_options.py 100000 1
  ;;
51) # the host-column entry points over random lengths and context counts, 400 s
  FUZZ_SECONDS=400 run 1000 fuzz_host.txt python tools/fuzz_host.py 100000 1
  ;;
52) # corrupted blobs: refused or decoded inside their buffers, 300 s
  FUZZ_SECONDS=300 run 600 fuzz_blob.txt python tools/fuzz_blob.py 1000000 1
  ;;
53) # more of the three fuzzers with new seeds (200 000 columns, 6 888 option sets, 4 545 host columns: no difference)
  ALPGPU_FUZZ_ROUNDS=80000 ALPGPU_FUZZ_SEED_BASE=9000000 run 1500 fuzz.txt python -m pytest tests/test_fuzz_gpu.py -m gpu -q -p no:cacheprovider -n 8
  FUZZ_SECONDS=900 run 1200 fuzz_options.txt python tools/fuzz_options.py 1000000 2
  FUZZ_SECONDS=500 run 800 fuzz_host.txt python tools/fuzz_host.py 1000000 2
  ;;
56) # third round of the fuzzers (9 889 option sets, 250 000 columns: no difference)
  FUZZ_SECONDS=1200 run 1500 fuzz_options.txt python tools/fuzz_options.py 1000000 3
  ALPGPU_FUZZ_ROUNDS=100000 ALPGPU_FUZZ_SEED_BASE=50000000 run 1200 fuzz.txt python -m pytest tests/test_fuzz_gpu.py -m gpu -q -p no:cacheprovider -n 8
  ;;
final) # the closing run: whole GPU suite, smoke, the bench line, configs[4] at N = 1 (tools/profile_round.sh r06 is a call of its own)
  run 900 tests.txt python -m pytest tests -m gpu -x -q
  run 300 smoke.txt python -c "import __graft_entry__ as g; g.smoke()"
  timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
  timeout 900 python bench.py --gpus 1 --column-gb 100 --steps 10 --warmup 3 > $out/bench_configs4_n1.json 2> $out/bench_configs4.err; echo "configs4 rc=$?"
  ;;
*) echo "unknown step $step";;
esac
tail -n 40 $out/*.txt | cut -c1-400
