#!/bin/bash
# round 4, GPU call 26: four narrow vectors per workgroup under residency caps (32 / 24 / 20 / 16 vectors in flight per CU) on 1..10 bits
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c26; mkdir -p $O
W=1,2,3,4,5,6,7,8,10
for i in 1 2; do
for pad in 0 10 14 20; do SWEEP_VPW=4 ALPGPU_DECODE_PAD_LDS_KIB=$pad timeout 300 python tools/sweep_residency.py $W 2>&1 | grep -v amdgpu.ids | sed 's/^/four /' | tee -a $O/four.txt; done
SWEEP_VPW=2 ALPGPU_DECODE_PAD_LDS_KIB=0 timeout 300 python tools/sweep_residency.py $W 2>&1 | grep -v amdgpu.ids | sed 's/^/two  /' | tee -a $O/four.txt
done
