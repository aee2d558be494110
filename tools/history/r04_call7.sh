#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c7; mkdir -p $O
for lib in alp_amd/libalpgpu.so build/variants/libalpgpu_prio0.so build/variants/libalpgpu_prio1.so alp_amd/libalpgpu.so build/variants/libalpgpu_prio0.so; do
  ALPGPU_LIB=$PWD/$lib ALPGPU_ENCODE_KERNEL=0 timeout 300 python tools/time_encode.py 1048576 mixed rd >> $O/time_encode.txt 2>&1
done
grep -v amdgpu.ids $O/time_encode.txt
