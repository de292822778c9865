mkdir -p gpurun_out/r05_exp5
timeout 900 python -m pytest tests/test_gpu_msm_midsize.py -x -q -k "four_records" > gpurun_out/r05_exp5/pytest.txt 2>&1
tail -5 gpurun_out/r05_exp5/pytest.txt
R5_STAGES=1 timeout 600 python tools/r05_ab_mid.py 16384 32768 65536 131072 > gpurun_out/r05_exp5/ab_mid.txt 2>&1
grep "##" gpurun_out/r05_exp5/ab_mid.txt
bash tools/r05_trace_mid.sh "reduce_batch=0" "reduce_batch=1" > gpurun_out/r05_exp5/trace.txt 2>&1
grep -h "k_digit_sums\|per MSM" gpurun_out/r05_trace_mid/cfg1.txt gpurun_out/r05_trace_mid/cfg2.txt | head
