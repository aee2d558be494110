#!/bin/bash
# round 4, GPU call 23: two vectors per workgroup with 8 / 7 / 6 workgroups per CU (16 / 14 / 12 vectors in flight) on 8..18 bits
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c23; mkdir -p $O
W=4,8,9,10,11,12,13,14,15,16,17,18
for i in 1 2; do for pad in 0 3 6; do SWEEP_VPW=2 ALPGPU_DECODE_PAD_LDS_KIB=$pad timeout 300 python tools/sweep_residency.py $W 2>&1 | grep -v amdgpu.ids | sed 's/^/two /' | tee -a $O/two.txt; done
SWEEP_VPW=1 ALPGPU_DECODE_PAD_LDS_KIB=0 timeout 300 python tools/sweep_residency.py $W 2>&1 | grep -v amdgpu.ids | sed 's/^/one /' | tee -a $O/two.txt; done
