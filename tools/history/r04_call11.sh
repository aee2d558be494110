#!/bin/bash
# round 4, GPU call 11: non-temporal loads (all wavefronts but the one that re-reads) and stores in the lean encode and the float encode, against the plain policy
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c11; mkdir -p $O
timeout 900 python -m pytest tests/test_encode_gpu.py tests/test_float_gpu.py tests/test_async_init_gpu.py tests/test_fuzz_gpu.py tests/test_recovery_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for i in 1 2 3; do
ALPGPU_LIB=build/variants/libalpgpu_plainpol.so timeout 300 python tools/time_encode.py 1048576 mixed rd 2>&1 | grep -v amdgpu.ids | tee -a $O/time_encode.txt
timeout 300 python tools/time_encode.py 1048576 mixed rd 2>&1 | grep -v amdgpu.ids | tee -a $O/time_encode.txt
done
for i in 1 2; do
ALPGPU_LIB=build/variants/libalpgpu_plainpol.so timeout 300 python tools/time_encode_f32.py 1048576 2>&1 | grep -v amdgpu.ids | tee -a $O/time_f32.txt
timeout 300 python tools/time_encode_f32.py 1048576 2>&1 | grep -v amdgpu.ids | tee -a $O/time_f32.txt
done
