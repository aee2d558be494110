#!/bin/bash
# round 4, GPU call 20: residency caps 8 / 7 / 6 / 5 per CU on 26..53 bits; wide-vector workgroups that idle before their loads (per-workgroup throttle) on the mix
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c20; mkdir -p $O
W=mix,26,28,30,32,34,36,40,48,53
for pad in 0 11 14 18; do ALPGPU_DECODE_PAD_LDS_KIB=$pad timeout 300 python tools/sweep_residency.py $W 2>&1 | grep -v amdgpu.ids | tee -a $O/residency.txt; done
for cfg in "34 2" "34 4" "34 6" "34 9" "28 3" "28 6" "40 4" "40 8"; do set -- $cfg
ALPGPU_DECODE_IDLE_FROM=$1 ALPGPU_DECODE_IDLE_UNITS=$2 timeout 300 python tools/sweep_residency.py mix,24,32,36,44,53 2>&1 | grep -v amdgpu.ids | tee -a $O/idle.txt; done
timeout 300 python tools/sweep_residency.py mix,24,32,36,44,53 2>&1 | grep -v amdgpu.ids | tee -a $O/idle.txt
