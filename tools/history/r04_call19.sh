#!/bin/bash
# round 4, GPU call 19: fewer decode workgroups resident per CU (unused dynamic LDS) on wide vectors
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c19; mkdir -p $O
for pad in 0 14 18 23 0 14 18 23; do ALPGPU_DECODE_PAD_LDS_KIB=$pad timeout 300 python tools/sweep_residency.py 2>&1 | grep -v amdgpu.ids | tee -a $O/residency.txt; done
