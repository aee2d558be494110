#!/bin/bash
# round 4, GPU call 18: float single pass at 6 / 7 / 8 wavefronts per SIMD (the scalar register file: 80 registers per wavefront at 8 — 89 spilled —, 102 at 6 — 18 spilled)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c18; mkdir -p $O
for i in 1 2 3; do
for v in f32occ6 f32occ7; do ALPGPU_LIB=build/variants/libalpgpu_$v.so timeout 300 python tools/time_encode_f32.py 1048576 2>&1 | grep -v amdgpu.ids | tee -a $O/time_f32.txt; done
timeout 300 python tools/time_encode_f32.py 1048576 2>&1 | grep -v amdgpu.ids | tee -a $O/time_f32.txt
done
ALPGPU_LIB=build/variants/libalpgpu_f32occ6.so timeout 200 tools/pmc_busy.sh f32occ6 python tools/prof_float.py 262144 > /dev/null 2>&1
grep "k_encode_fused_f32<0>" gpurun_out/pmcb_f32occ6.txt | cut -c1-420 | tee $O/pmc.txt
