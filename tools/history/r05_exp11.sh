mkdir -p gpurun_out/r05_exp11
timeout 1200 python -m pytest tests/test_gpu_prove_batch.py tests/test_gpu_ipa.py tests/test_gpu_batch_dev.py tests/test_gpu_validate_points.py -x -q > gpurun_out/r05_exp11/pytest.txt 2>&1
tail -6 gpurun_out/r05_exp11/pytest.txt
python tools/bench_prove_batch.py 10 12 14 16 > gpurun_out/r05_exp11/prove_bench.txt 2>&1; cat gpurun_out/r05_exp11/prove_bench.txt
