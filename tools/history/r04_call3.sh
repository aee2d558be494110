#!/bin/bash
# round 4, GPU call 3: lean kernel after the pack diet (parity, timing A/B, counters) + the last-register probe over every mixed-width instruction
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c3; mkdir -p $O
T="tests/test_encode_gpu.py tests/test_fuzz_gpu.py tests/test_recovery_gpu.py tests/test_reference_gpu.py tests/test_async_init_gpu.py"
timeout 1200 python -m pytest $T -x -q -m gpu > $O/tests_lean.log 2>&1; echo "lean tests rc=$?"; tail -3 $O/tests_lean.log
for k in 0 1 0 1; do
  ALPGPU_ENCODE_KERNEL=$k timeout 300 python tools/time_encode.py 1048576 mixed rd >> $O/time_encode.txt 2>&1
done
grep -v amdgpu.ids $O/time_encode.txt
ALPGPU_ENCODE_KERNEL=0 timeout 300 tools/pmc_busy.sh r04lean2_mixed python tools/prof_encode.py mixed 262144 > /dev/null 2>&1
ALPGPU_ENCODE_KERNEL=0 timeout 300 tools/pmc_busy.sh r04lean2_rd python tools/prof_encode.py rd 262144 > /dev/null 2>&1
grep "k_encode_lean" gpurun_out/pmcb_r04lean2_mixed.txt gpurun_out/pmcb_r04lean2_rd.txt | cut -c1-360
( timeout 120 tools/last_vgpr_probe2 2048 20; timeout 120 tools/last_vgpr_probe2_margin 2048 20; timeout 120 tools/last_vgpr_probe2 6000 40 ) > $O/probe2.txt 2>&1
cat $O/probe2.txt
