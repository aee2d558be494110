#!/bin/bash
# round 4, GPU call 9: k_decode_pairs (workgroups that choose how to run their two vectors), the adaptive number of search workgroups per CU,
# the analysis without its third compare; fuzz on fresh seeds on this build
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c9; mkdir -p $O
timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_encode_gpu.py tests/test_async_init_gpu.py tests/test_fuzz_gpu.py tests/test_recovery_gpu.py tests/test_reference_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout 600 python tools/sweep_pairing.py 2>&1 | grep -v amdgpu.ids | tee $O/sweep_pairing.txt
for i in 1 2; do
ALPGPU_ASYNC_INIT_ADAPTIVE=0 timeout 300 python tools/time_encode.py 1048576 mixed rd 2>&1 | grep -v amdgpu.ids | sed 's/^/adaptive=0 /' | tee -a $O/time_encode.txt
ALPGPU_ASYNC_INIT_ADAPTIVE=1 timeout 300 python tools/time_encode.py 1048576 mixed rd 2>&1 | grep -v amdgpu.ids | sed 's/^/adaptive=1 /' | tee -a $O/time_encode.txt
done
ALPGPU_ENCODE_KERNEL=1 timeout 300 python tools/time_encode.py 1048576 mixed rd 2>&1 | grep -v amdgpu.ids | sed 's/^/classic /' | tee -a $O/time_encode.txt
ALPGPU_FUZZ_ROUNDS=3000 ALPGPU_FUZZ_SEED_BASE=9000000 timeout 1200 python -m pytest tests/test_fuzz_gpu.py -x -q -m gpu > $O/fuzz_lean.log 2>&1; echo "fuzz lean rc=$?"; tail -2 $O/fuzz_lean.log
ALPGPU_ENCODE_KERNEL=1 ALPGPU_FUZZ_ROUNDS=1000 ALPGPU_FUZZ_SEED_BASE=9500000 timeout 900 python -m pytest tests/test_fuzz_gpu.py -x -q -m gpu > $O/fuzz_classic.log 2>&1; echo "fuzz classic rc=$?"; tail -2 $O/fuzz_classic.log
