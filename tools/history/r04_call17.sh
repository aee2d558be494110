#!/bin/bash
# round 4, GPU call 17: where the float single pass's instructions go (-DALPGPU_F32_STOP_AT=n builds, counters per wavefront, bench.py's decimal_mixed float column)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c17; mkdir -p $O
export ALPGPU_PROF_ENCODE_ONLY=1
for n in 1 2 3 4 5 6 7 8; do
ALPGPU_LIB=build/variants/libalpgpu_fstop$n.so timeout 200 tools/pmc_busy.sh fstop$n python tools/prof_float.py 262144 > /dev/null 2>&1
done
timeout 200 tools/pmc_busy.sh ffull python tools/prof_float.py 262144 > /dev/null 2>&1
for n in fstop1 fstop2 fstop3 fstop4 fstop5 fstop6 fstop7 fstop8 ffull; do echo -n "$n: "; grep "k_encode_fused_f32<0>" gpurun_out/pmcb_$n.txt | cut -c1-400; done | tee $O/stages_f32.txt
