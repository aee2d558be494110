#!/bin/bash
# round 4, GPU call 14: the look-back's first round asked for in front of the exception record's stage (default) / in front of the pack / behind the record
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c14; mkdir -p $O
timeout 900 python -m pytest tests/test_encode_gpu.py tests/test_async_init_gpu.py tests/test_fuzz_gpu.py tests/test_recovery_gpu.py tests/test_reference_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for i in 1 2 3; do
for v in nolook lookearly; do ALPGPU_LIB=build/variants/libalpgpu_$v.so timeout 300 python tools/time_encode.py 1048576 mixed rd 2>&1 | grep -v amdgpu.ids | tee -a $O/time_encode.txt; done
timeout 300 python tools/time_encode.py 1048576 mixed rd 2>&1 | grep -v amdgpu.ids | tee -a $O/time_encode.txt
done
