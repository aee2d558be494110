#!/bin/bash
# round 4, GPU call 12: where the lean encode's instructions go — builds that end every wavefront behind stage n (-DALPGPU_LEAN_STOP_AT=n), counters per wavefront
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c12; mkdir -p $O
export ALPGPU_PROF_ENCODE_ONLY=1
for n in 1 2 3 4 5 6 7 8; do
ALPGPU_LIB=build/variants/libalpgpu_stop$n.so timeout 200 tools/pmc_busy.sh stop${n}_mixed python tools/prof_encode.py mixed 262144 > /dev/null 2>&1
ALPGPU_LIB=build/variants/libalpgpu_stop$n.so timeout 200 tools/pmc_busy.sh stop${n}_rd python tools/prof_encode.py rd 262144 > /dev/null 2>&1
done
timeout 200 tools/pmc_busy.sh full_mixed python tools/prof_encode.py mixed 262144 > /dev/null 2>&1
timeout 200 tools/pmc_busy.sh full_rd python tools/prof_encode.py rd 262144 > /dev/null 2>&1
for t in mixed rd; do for n in stop1 stop2 stop3 stop4 stop5 stop6 stop7 stop8 full; do echo -n "$n $t: "; grep k_encode_lean gpurun_out/pmcb_${n}_$t.txt | cut -c1-400; done; done | tee $O/stages.txt
