#!/bin/bash
# round 4, GPU call 24: with 20 exceptions per vector: two vectors per workgroup / the pair kernel under residency caps, against one vector per workgroup
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c24; mkdir -p $O
export SWEEP_EXC=20
W=2,4,8,10,12,14,16,18,20,22,24,28
for pad in 0 3 6; do SWEEP_VPW=2 ALPGPU_DECODE_PAD_LDS_KIB=$pad timeout 300 python tools/sweep_residency.py $W 2>&1 | grep -v amdgpu.ids | sed 's/^/two   /' | tee -a $O/exc.txt; done
for pad in 0 3 6; do SWEEP_VPW=2 SWEEP_PAIRING=1 ALPGPU_DECODE_PAD_LDS_KIB=$pad timeout 300 python tools/sweep_residency.py $W 2>&1 | grep -v amdgpu.ids | sed 's/^/pairs /' | tee -a $O/exc.txt; done
for pad in 0 11; do SWEEP_VPW=1 ALPGPU_DECODE_PAD_LDS_KIB=$pad timeout 300 python tools/sweep_residency.py $W 2>&1 | grep -v amdgpu.ids | sed 's/^/one   /' | tee -a $O/exc.txt; done
unset SWEEP_EXC
W=1,2,3,4,5,6,7,8,9,10
for pad in 0 3; do SWEEP_VPW=2 ALPGPU_DECODE_PAD_LDS_KIB=$pad timeout 300 python tools/sweep_residency.py $W 2>&1 | grep -v amdgpu.ids | sed 's/^/two noexc /' | tee -a $O/narrow.txt; done
