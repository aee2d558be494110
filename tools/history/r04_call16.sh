#!/bin/bash
# round 4, GPU call 16: float single pass with late kernel arguments; the pair kernel under the auto rule (narrow columns with exceptions)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c16; mkdir -p $O
timeout 900 python -m pytest tests/test_float_gpu.py tests/test_decode_gpu.py tests/test_fuzz_gpu.py tests/test_recovery_gpu.py tests/test_reference_gpu.py tests/test_encode_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for i in 1 2 3; do
ALPGPU_LIB=build/variants/libalpgpu_f32entry.so timeout 300 python tools/time_encode_f32.py 1048576 2>&1 | grep -v amdgpu.ids | tee -a $O/time_f32.txt
timeout 300 python tools/time_encode_f32.py 1048576 2>&1 | grep -v amdgpu.ids | tee -a $O/time_f32.txt
done
timeout 200 tools/pmc_busy.sh f32late python tools/prof_float.py 262144 > /dev/null 2>&1
ALPGPU_LIB=build/variants/libalpgpu_f32entry.so timeout 200 tools/pmc_busy.sh f32entry python tools/prof_float.py 262144 > /dev/null 2>&1
grep "k_encode_fused_f32<0>" gpurun_out/pmcb_f32late.txt gpurun_out/pmcb_f32entry.txt | cut -c1-420 | tee $O/pmc.txt
