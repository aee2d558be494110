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
*) echo "unknown step $step";;
esac
tail -n 40 $out/*.txt | cut -c1-400
