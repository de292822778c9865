#!/bin/bash
# one GPU call: the new tests, option A/B (direct_result), fuzz runs
out=${1:-gpurun_out/r04g}
mkdir -p "$out"
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_ipa.py tests/test_gpu_abi_errors.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -6
timeout 1500 python -m pytest tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -6
for o in "direct_result=0" "direct_result=1" "direct_result=0" "direct_result=1"; do
  python tools/r04_msm_latency.py $o
done 2>&1 | tee "$out/direct_result_ab.txt"
timeout 200 python tools/fuzz_msm.py 120 2>&1 | tail -3 | tee "$out/fuzz_msm.txt"
timeout 200 python tools/fuzz_ops.py 100 2>&1 | tail -3 | tee "$out/fuzz_ops.txt"
