#!/bin/bash
# round 4, GPU call 34 (AS IT RAN; the knob ALPGPU_SIDE_SEARCH_PLAIN no longer exists): the search beside the encode as plain launches on the side stream instead of the persistent kernel — the encode starves them, 8 s per call through the recovery route (profiles/r04_encode_levers.txt point 8i)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c34; mkdir -p $O
for chunk in 0 100000 2048 512; do
echo "--- ALPGPU_SIDE_SEARCH_PLAIN=$chunk" | tee -a $O/plain.txt
ALPGPU_SIDE_SEARCH_PLAIN=$chunk timeout 300 python tools/time_encode_f32.py 1048576 2>&1 | grep -v amdgpu.ids | tee -a $O/plain.txt
ALPGPU_SIDE_SEARCH_PLAIN=$chunk timeout 300 python tools/time_encode.py 1048576 mixed rd 2>&1 | grep -v amdgpu.ids | tee -a $O/plain.txt
done
for prio in 1 -1; do
echo "--- ALPGPU_SIDE_SEARCH_PLAIN=2048 ALPGPU_INIT_STREAM_PRIO=$prio" | tee -a $O/plain.txt
ALPGPU_INIT_STREAM_PRIO=$prio ALPGPU_SIDE_SEARCH_PLAIN=2048 timeout 300 python tools/time_encode_f32.py 1048576 2>&1 | grep -v amdgpu.ids | tee -a $O/plain.txt
done
