#!/bin/bash
# round 4, GPU call 6: single validated state read; whole GPU suite; clocks under load
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c6; mkdir -p $O
for k in 0 1 0 1; do
  ALPGPU_ENCODE_KERNEL=$k timeout 300 python tools/time_encode.py 1048576 mixed rd >> $O/time_encode.txt 2>&1
done
grep -v amdgpu.ids $O/time_encode.txt
ALPGPU_ENCODE_KERNEL=0 timeout 120 python tools/clock_under_load.py mixed 4 2>&1 | grep -v amdgpu.ids | tee $O/clock_mixed.txt
timeout 120 python tools/clock_under_load.py decode 4 2>&1 | grep -v amdgpu.ids | tee $O/clock_decode.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests_all.log 2>&1; echo "all gpu tests rc=$?"; tail -5 $O/tests_all.log
