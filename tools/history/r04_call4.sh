#!/bin/bash
# round 4, GPU call 4: lean kernel next to an 8-wavefront search workgroup (tile-shaped); where the lean kernel's instructions go (ablation counters)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c4; mkdir -p $O
for lib in build/variants/libalpgpu_s8.so alp_amd/libalpgpu.so build/variants/libalpgpu_s8.so; do
  ALPGPU_LIB=$PWD/$lib ALPGPU_ENCODE_KERNEL=0 timeout 300 python tools/time_encode.py 1048576 mixed rd >> $O/time_encode.txt 2>&1
done
ALPGPU_LIB=$PWD/build/variants/libalpgpu_s8.so ALPGPU_ENCODE_KERNEL=1 timeout 300 python tools/time_encode.py 1048576 mixed rd >> $O/time_encode.txt 2>&1
grep -v amdgpu.ids $O/time_encode.txt
ALPGPU_LIB=$PWD/build/variants/libalpgpu_s8.so timeout 200 python -m pytest tests/test_async_init_gpu.py -x -q -m gpu 2>&1 | tail -2
for v in nosecond nopack noexc; do
  ALPGPU_LIB=$PWD/build/variants/libalpgpu_$v.so ALPGPU_ENCODE_KERNEL=0 timeout 300 tools/pmc_busy.sh r04abl_$v python tools/prof_encode.py mixed 262144 > /dev/null 2>&1
  grep "k_encode_lean" gpurun_out/pmcb_r04abl_$v.txt | cut -c1-360
done
