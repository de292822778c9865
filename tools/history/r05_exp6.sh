mkdir -p gpurun_out/r05_exp6
timeout 1200 python -m pytest tests/test_gpu_prove_batch.py -x -q > gpurun_out/r05_exp6/pytest_prove.txt 2>&1
tail -15 gpurun_out/r05_exp6/pytest_prove.txt
timeout 900 python -m pytest tests/test_gpu_msm_midsize.py -x -q > gpurun_out/r05_exp6/pytest_midsize.txt 2>&1
tail -5 gpurun_out/r05_exp6/pytest_midsize.txt
