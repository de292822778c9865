mkdir -p gpurun_out/r05_exp10
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r05_exp10/pytest_gpu_all.txt 2>&1
tail -8 gpurun_out/r05_exp10/pytest_gpu_all.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_exp10/bench_default.json 2> gpurun_out/r05_exp10/bench_default.err
tail -3 gpurun_out/r05_exp10/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_exp10/bench_default.json').read().strip().splitlines()[-1])
e=d['extra']
print('headline', d['value'], d['ms_per_step'])
c2=e['C2_msm_2e16']; print('C2', c2['ms_per_msm_one_at_a_time'], c2['ms_per_msm_two_in_flight'])
c5=e['C5_batch_verify']; print('C5', c5['value'], c5['batch_latency_s'], c5['wire_format_2']['batch_latency_s'], c5['batch_prover']['proves_per_s'])
c3=e['C3_ipa_prover']; print('C3', c3['value'], c3.get('with_fixed_generators'))
c4=e['C4_aggregated_range_proof']; print('C4', c4['value'])
PY
