#!/bin/bash
# round 4, GPU call 15: a wavefront touches the lines of the vector its successor-by-one-generation will load (prefetch into the memory-side cache)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c15; mkdir -p $O
for i in 1 2 3; do
for v in pf384 pf768 pf1536; do ALPGPU_LIB=build/variants/libalpgpu_$v.so timeout 300 python tools/time_encode.py 1048576 mixed rd 2>&1 | grep -v amdgpu.ids | tee -a $O/time_encode.txt; done
timeout 300 python tools/time_encode.py 1048576 mixed rd 2>&1 | grep -v amdgpu.ids | tee -a $O/time_encode.txt
done
