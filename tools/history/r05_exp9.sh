mkdir -p gpurun_out/r05_exp9
timeout 1200 python -m pytest tests/test_gpu_batch_dev.py tests/test_gpu_configs.py tests/test_gpu_msm.py tests/test_gpu_ipa.py tests/test_gpu_dist.py -x -q > gpurun_out/r05_exp9/pytest.txt 2>&1
tail -8 gpurun_out/r05_exp9/pytest.txt
for d in 2 3 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-extra --soak-seconds 0 --steps 200 --warmup 20 --depth $d 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('depth', $d, 'ms_per_step', d['ms_per_step'], 'value', d['value'])"; done > gpurun_out/r05_exp9/depth_ab.txt 2>&1
cat gpurun_out/r05_exp9/depth_ab.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_exp9/bench_default.json 2> gpurun_out/r05_exp9/bench_default.err
tail -c 6000 gpurun_out/r05_exp9/bench_default.json; tail -5 gpurun_out/r05_exp9/bench_default.err
