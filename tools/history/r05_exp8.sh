mkdir -p gpurun_out/r05_exp8
timeout 900 python -m pytest tests/test_gpu_validate_points.py -x -q > gpurun_out/r05_exp8/pytest_validate.txt 2>&1
tail -15 gpurun_out/r05_exp8/pytest_validate.txt
timeout 600 python tools/r05_validate_cost.py > gpurun_out/r05_exp8/validate_cost.txt 2>&1
tail -14 gpurun_out/r05_exp8/validate_cost.txt
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r05_exp8/pytest_gpu_all.txt 2>&1
tail -8 gpurun_out/r05_exp8/pytest_gpu_all.txt
