#!/bin/bash
# round 4, GPU call 2: the lean encode kernel — parity (both kernels), timing A/B against the classic kernel, counters
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c2; mkdir -p $O
T="tests/test_encode_gpu.py tests/test_fuzz_gpu.py tests/test_recovery_gpu.py tests/test_reference_gpu.py tests/test_async_init_gpu.py tests/test_sharding_gpu.py tests/test_container_gpu.py"
timeout 1200 python -m pytest $T -x -q -m gpu > $O/tests_lean.log 2>&1; echo "lean tests rc=$?"; tail -4 $O/tests_lean.log
ALPGPU_ENCODE_KERNEL=1 timeout 1200 python -m pytest tests/test_encode_gpu.py tests/test_fuzz_gpu.py tests/test_recovery_gpu.py tests/test_async_init_gpu.py -x -q -m gpu > $O/tests_classic.log 2>&1; echo "classic tests rc=$?"; tail -3 $O/tests_classic.log
for k in 0 1 0 1; do
  ALPGPU_ENCODE_KERNEL=$k timeout 300 python tools/time_encode.py 1048576 mixed rd >> $O/time_encode.txt 2>&1
done
grep -v amdgpu.ids $O/time_encode.txt
ALPGPU_ENCODE_KERNEL=0 timeout 300 tools/pmc_busy.sh r04lean_mixed python tools/prof_encode.py mixed 262144 > /dev/null 2>&1
ALPGPU_ENCODE_KERNEL=0 timeout 300 tools/pmc_busy.sh r04lean_rd python tools/prof_encode.py rd 262144 > /dev/null 2>&1
grep "k_encode_lean\|k_rowgroup_init" gpurun_out/pmcb_r04lean_mixed.txt gpurun_out/pmcb_r04lean_rd.txt | cut -c1-360
