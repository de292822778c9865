mkdir -p gpurun_out/r05_exp2
timeout 900 python -m pytest tests/test_gpu_msm_midsize.py -x -q > gpurun_out/r05_exp2/pytest_midsize.txt 2>&1
tail -5 gpurun_out/r05_exp2/pytest_midsize.txt
R5_STAGES=1 timeout 600 python tools/r05_ab_mid.py > gpurun_out/r05_exp2/ab_mid.txt 2>&1
grep "##" gpurun_out/r05_exp2/ab_mid.txt
