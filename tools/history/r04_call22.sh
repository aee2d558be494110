#!/bin/bash
# round 4, GPU call 22: the refined residency rule (double), the same experiment on the float decode
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c22; mkdir -p $O
timeout 600 python -m pytest tests/test_decode_gpu.py tests/test_float_gpu.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do for pad in 0 11 14 18; do ALPGPU_DECODE_F32_PAD_LDS_KIB=$pad timeout 300 python tools/time_decode_f32.py 2>&1 | grep -v amdgpu.ids | tee -a $O/f32.txt; done; done
ALPGPU_DECODE_PAD_LDS_KIB=0 timeout 300 python tools/sweep_residency_rule.py 2>&1 | grep -v amdgpu.ids | tee -a $O/rule.txt
timeout 300 python tools/sweep_residency_rule.py 2>&1 | grep -v amdgpu.ids | tee -a $O/rule.txt
for i in 1 2; do
ALPGPU_DECODE_PAD_LDS_KIB=0 timeout 300 python tools/prof_encode.py rd 1048576 2>&1 | grep "decode of" | sed 's/^/no cap /' | tee -a $O/rd.txt
timeout 300 python tools/prof_encode.py rd 1048576 2>&1 | grep "decode of" | sed 's/^/rule   /' | tee -a $O/rd.txt
done
