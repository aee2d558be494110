#!/bin/bash
# round 4, GPU call 13: the lean encode's late kernel arguments (read from the kernarg segment where they are used) against arguments at the entry
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c13; mkdir -p $O
timeout 900 python -m pytest tests/test_encode_gpu.py tests/test_async_init_gpu.py tests/test_fuzz_gpu.py tests/test_recovery_gpu.py tests/test_reference_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for i in 1 2 3; do
ALPGPU_LIB=build/variants/libalpgpu_argsentry.so timeout 300 python tools/time_encode.py 1048576 mixed rd 2>&1 | grep -v amdgpu.ids | tee -a $O/time_encode.txt
timeout 300 python tools/time_encode.py 1048576 mixed rd 2>&1 | grep -v amdgpu.ids | tee -a $O/time_encode.txt
done
timeout 200 tools/pmc_busy.sh late_mixed python tools/prof_encode.py mixed 262144 > /dev/null 2>&1
ALPGPU_LIB=build/variants/libalpgpu_argsentry.so timeout 200 tools/pmc_busy.sh entry_mixed python tools/prof_encode.py mixed 262144 > /dev/null 2>&1
grep k_encode_lean gpurun_out/pmcb_late_mixed.txt gpurun_out/pmcb_entry_mixed.txt | cut -c1-420 | tee $O/pmc.txt
