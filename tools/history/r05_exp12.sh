mkdir -p gpurun_out/r05_exp12
timeout 1200 python -m pytest tests/test_gpu_prove_batch.py -x -q > gpurun_out/r05_exp12/pytest.txt 2>&1
tail -6 gpurun_out/r05_exp12/pytest.txt
for tw in 8 10 11 12 13 8; do PB_TW=$tw python tools/bench_prove_batch.py 12 14 2>&1; done > gpurun_out/r05_exp12/prove_tw.txt
cat gpurun_out/r05_exp12/prove_tw.txt | cut -c1-420
