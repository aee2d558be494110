#!/bin/bash
# round 4, GPU call 21: the residency rule of the one-vector-per-workgroup decode (six / seven workgroups per CU on wide columns) against no cap; ALP_RD column decode
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c21; mkdir -p $O
timeout 600 python -m pytest tests/test_decode_gpu.py tests/test_sharding_gpu.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do
ALPGPU_DECODE_PAD_LDS_KIB=0 timeout 300 python tools/sweep_residency_rule.py 2>&1 | grep -v amdgpu.ids | tee -a $O/rule.txt
timeout 300 python tools/sweep_residency_rule.py 2>&1 | grep -v amdgpu.ids | tee -a $O/rule.txt
done
for i in 1 2; do
ALPGPU_DECODE_PAD_LDS_KIB=0 timeout 300 python tools/prof_encode.py rd 1048576 2>&1 | grep "decode of" | sed 's/^/no cap /' | tee -a $O/rd.txt
timeout 300 python tools/prof_encode.py rd 1048576 2>&1 | grep "decode of" | sed 's/^/rule   /' | tee -a $O/rd.txt
ALPGPU_DECODE_PAD_LDS_KIB=11 timeout 300 python tools/prof_encode.py rd 1048576 2>&1 | grep "decode of" | sed 's/^/seven  /' | tee -a $O/rd.txt
ALPGPU_DECODE_PAD_LDS_KIB=18 timeout 300 python tools/prof_encode.py rd 1048576 2>&1 | grep "decode of" | sed 's/^/five   /' | tee -a $O/rd.txt
done
