#!/bin/bash
# round 4, GPU call 8: float single pass with the image left in LDS (62 VGPRs, four tiles per CU), state poll in front, barrier; tile-shaped float search beside it
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c8; mkdir -p $O
timeout 900 python -m pytest tests/test_float_gpu.py tests/test_fuzz_gpu.py tests/test_recovery_gpu.py tests/test_reference_gpu.py tests/test_async_init_gpu.py tests/test_encode_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout 300 python tools/time_encode_f32.py 1048576 2>&1 | grep -v amdgpu.ids | tee $O/time_f32.txt
timeout 300 python tools/time_encode_f32.py 1048576 2>&1 | grep -v amdgpu.ids | tee -a $O/time_f32.txt
ALPGPU_ENCODE_KERNEL=0 timeout 300 python tools/time_encode.py 1048576 mixed rd 2>&1 | grep -v amdgpu.ids | tee $O/time_f64.txt
