#!/bin/bash
# round 4, GPU call 32: the lean tile's LDS init barrier behind the issue of the loads (default) against in front of them
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c32; mkdir -p $O
timeout 900 python -m pytest tests/test_encode_gpu.py tests/test_async_init_gpu.py tests/test_fuzz_gpu.py tests/test_recovery_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
for i in 1 2 3; do
ALPGPU_LIB=build/variants/libalpgpu_initfirst.so timeout 300 python tools/time_encode.py 1048576 mixed rd 2>&1 | grep -v amdgpu.ids | tee -a $O/time_encode.txt
timeout 300 python tools/time_encode.py 1048576 mixed rd 2>&1 | grep -v amdgpu.ids | tee -a $O/time_encode.txt
done
