#!/bin/bash
# the bench lines of the committed library against the committed profiles (run after tools/profile_round.sh + summarize_round.py)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04final; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 900 python bench.py --gpus 1 --column-gb 100 --steps 10 --warmup 3 > $O/bench_configs4_n1.json 2> $O/bench_configs4.err; echo "configs4 rc=$?"
timeout 300 python tools/time_sink_f32.py 1048576 2>&1 | grep -v amdgpu.ids | tee $O/time_sink_f32.txt
for t in mixed rd; do timeout 300 tools/pmc_busy.sh r04final_$t python tools/prof_encode.py $t 262144 > /dev/null 2>&1; done
ALPGPU_ENCODE_KERNEL=1 timeout 300 tools/pmc_busy.sh r04final_mixed_classic python tools/prof_encode.py mixed 262144 > /dev/null 2>&1
timeout 300 tools/pmc_busy.sh r04final_f32 python tools/prof_float.py 262144 > /dev/null 2>&1
timeout 300 tools/pmc_busy.sh r04final_sinkf python tools/prof_sink_direct_f32.py 262144 > /dev/null 2>&1
cat gpurun_out/pmcb_r04final_*.txt | grep -v "k_encode_analyze\|k_scan\|k_encode_pack\|k_fused_finish\|k_count_rd" | cut -c1-360 > $O/pmc_busy.txt
