#!/bin/bash
# round 4, closing GPU call: whole GPU suite, the round's profile (kernel stats + HBM traffic, tools/profile_round.sh), the bench line, configs[4] at N = 1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04final; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests_all.log 2>&1; echo "all gpu tests rc=$?"; tail -3 $O/tests_all.log
timeout 1500 bash tools/profile_round.sh r04 > $O/profile_round.log 2>&1; echo "profile rc=$?"; tail -5 $O/profile_round.log | cut -c1-300
cp -r gpurun_out/r04_prof/*.json $O/ 2>/dev/null
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 900 python bench.py --gpus 1 --column-gb 100 --steps 10 --warmup 3 > $O/bench_configs4_n1.json 2> $O/bench_configs4.err; echo "configs4 rc=$?"
ALPGPU_ENCODE_KERNEL=0 timeout 300 python tools/time_encode.py 1048576 mixed rd 2>&1 | grep -v amdgpu.ids | tee $O/time_encode_lean.txt
ALPGPU_ENCODE_KERNEL=1 timeout 300 python tools/time_encode.py 1048576 mixed rd 2>&1 | grep -v amdgpu.ids | tee $O/time_encode_classic.txt
timeout 300 python tools/time_encode_f32.py 1048576 2>&1 | grep -v amdgpu.ids | tee $O/time_encode_f32.txt
timeout 300 python tools/time_sink_f32.py 1048576 2>&1 | grep -v amdgpu.ids | tee $O/time_sink_f32.txt
ALPGPU_ENCODE_KERNEL=0 timeout 300 tools/pmc_busy.sh r04final_mixed python tools/prof_encode.py mixed 262144 > /dev/null 2>&1
ALPGPU_ENCODE_KERNEL=0 timeout 300 tools/pmc_busy.sh r04final_rd python tools/prof_encode.py rd 262144 > /dev/null 2>&1
ALPGPU_ENCODE_KERNEL=1 timeout 300 tools/pmc_busy.sh r04final_mixed_classic python tools/prof_encode.py mixed 262144 > /dev/null 2>&1
timeout 300 tools/pmc_busy.sh r04final_f32 python tools/prof_float.py 262144 > /dev/null 2>&1
timeout 300 tools/pmc_busy.sh r04final_sinkf python tools/prof_sink_direct_f32.py 262144 > /dev/null 2>&1
cat gpurun_out/pmcb_r04final_*.txt | grep -v "k_encode_analyze\|k_scan\|k_encode_pack\|k_fused_finish\|k_count_rd" | cut -c1-360
