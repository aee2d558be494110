#!/bin/bash
# round 4, GPU call 10: cache-policy and group-size variants of the lean encode; the float search as a persistent kernel in front; the decode crossover
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c10; mkdir -p $O
for i in 1 2; do for v in base ntst ntld ntboth grp1 grp4; do
ALPGPU_LIB=build/variants/libalpgpu_$v.so timeout 300 python tools/time_encode.py 1048576 mixed rd 2>&1 | grep -v amdgpu.ids | tee -a $O/time_encode_variants.txt
done; done
echo "--- float: default" | tee -a $O/time_f32.txt
timeout 300 python tools/time_encode_f32.py 1048576 2>&1 | grep -v amdgpu.ids | tee -a $O/time_f32.txt
for cfg in "1 4" "1 3" "0 5" "0 4"; do set -- $cfg
echo "--- float: persistent search IN FRONT (ALPGPU_ASYNC_SERIAL), tile-shaped=$1, workgroups per CU=$2 (column 'beside')" | tee -a $O/time_f32.txt
if [ $1 = 1 ]; then export ALPGPU_F32_SEARCH_TILE=1; else unset ALPGPU_F32_SEARCH_TILE; fi
ALPGPU_ASYNC_SERIAL=1 ALPGPU_ASYNC_INIT_WG_PER_CU=$2 timeout 300 python tools/time_encode_f32.py 1048576 2>&1 | grep -v amdgpu.ids | tee -a $O/time_f32.txt
done
unset ALPGPU_F32_SEARCH_TILE
timeout 600 python tools/sweep_crossover.py 2>&1 | grep -v amdgpu.ids | tee $O/crossover.txt
