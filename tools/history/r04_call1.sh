#!/bin/bash
# round 4, GPU call 1: changed tests, the bench with the full sweep, baseline encode timings (A/B of the state re-read), PMC of the encode kernels
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c1; mkdir -p $O
timeout 900 python -m pytest tests/test_host_pipeline_gpu.py tests/test_dropin_gpu.py tests/test_encode_gpu.py tests/test_bench_gpu.py tests/test_sharding_gpu.py tests/test_async_init_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -5 $O/tests.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
for lib in alp_amd/libalpgpu.so build/variants/libalpgpu_singleread.so alp_amd/libalpgpu.so build/variants/libalpgpu_singleread.so; do
  ALPGPU_LIB=$PWD/$lib timeout 300 python tools/time_encode.py 1048576 mixed rd >> $O/time_encode.txt 2>&1
done
cat $O/time_encode.txt
timeout 300 tools/pmc_busy.sh r04base_mixed python tools/prof_encode.py mixed 262144 > /dev/null 2>&1
timeout 300 tools/pmc_busy.sh r04base_rd python tools/prof_encode.py rd 262144 > /dev/null 2>&1
cat gpurun_out/pmcb_r04base_mixed.txt gpurun_out/pmcb_r04base_rd.txt | cut -c1-330
