#!/bin/bash
# round 4, GPU call 5: workers parked at a barrier (default) against spinning on an LDS word; tile-shaped search chosen at run time
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c5; mkdir -p $O
timeout 900 python -m pytest tests/test_encode_gpu.py tests/test_fuzz_gpu.py tests/test_recovery_gpu.py tests/test_async_init_gpu.py tests/test_reference_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for lib in alp_amd/libalpgpu.so build/variants/libalpgpu_spin2.so build/variants/libalpgpu_spin16.so alp_amd/libalpgpu.so build/variants/libalpgpu_spin2.so; do
  ALPGPU_LIB=$PWD/$lib ALPGPU_ENCODE_KERNEL=0 timeout 300 python tools/time_encode.py 1048576 mixed rd >> $O/time_encode.txt 2>&1
done
ALPGPU_ENCODE_KERNEL=1 timeout 300 python tools/time_encode.py 1048576 mixed rd >> $O/time_encode.txt 2>&1
grep -v amdgpu.ids $O/time_encode.txt
ALPGPU_ENCODE_KERNEL=0 timeout 300 tools/pmc_busy.sh r04lean3_mixed python tools/prof_encode.py mixed 262144 > /dev/null 2>&1
grep "k_encode_lean\|k_rowgroup" gpurun_out/pmcb_r04lean3_mixed.txt | cut -c1-360
