#!/bin/bash
# round 4, GPU call 25: the benchmark column (all widths by rowgroup) under two vectors per workgroup / the pair kernel with residency caps; the new rule on single widths
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c25; mkdir -p $O
timeout 600 python -m pytest tests/test_decode_gpu.py -x -q -m gpu 2>&1 | tail -1
for pad in 0 3 6 9 13; do SWEEP_VPW=2 ALPGPU_DECODE_PAD_LDS_KIB=$pad timeout 300 python tools/sweep_residency.py mix 2>&1 | grep -v amdgpu.ids | sed 's/^/two   /' | tee -a $O/mix.txt; done
for pad in 0 3 6 9; do SWEEP_VPW=2 SWEEP_PAIRING=1 ALPGPU_DECODE_PAD_LDS_KIB=$pad timeout 300 python tools/sweep_residency.py mix 2>&1 | grep -v amdgpu.ids | sed 's/^/pairs /' | tee -a $O/mix.txt; done
SWEEP_VPW=1 timeout 300 python tools/sweep_residency.py mix 2>&1 | grep -v amdgpu.ids | sed 's/^/one   /' | tee -a $O/mix.txt
SWEEP_VPW=0 timeout 300 python tools/sweep_residency.py 1,4,8,9,10,11,12,13,14,15,16,17,18,24,34,40,53 2>&1 | grep -v amdgpu.ids | sed 's/^/auto  /' | tee -a $O/mix.txt
