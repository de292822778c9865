#!/bin/bash
# the whole bench under the mixed-width table, with and without the one-launch kernel for PAIRS of mid-size MSMs (the inner-product rounds)
for cfg in "" "--opt mid_min=-1" "--opt mixed_windows=0" ""; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $cfg > gpurun_out/_b.json 2>/dev/null
  python3 -c "
import json
d=json.loads(open('gpurun_out/_b.json').read().strip().splitlines()[-1]); ex=d['extra']; c5=ex['C5_batch_verify']
print('cfg [$cfg]: %.4f ms/step | C2 %.4f / %.4f | C3 %.5f (fixed %.5f) | C4 prove %.5f | C5 %.4g (v2 %.4g; one at a time %.3f / %.3f ms) | prover %.4g' % (d['ms_per_step'], ex['C2_msm_2e16']['ms_per_msm_one_at_a_time'], ex['C2_msm_2e16']['ms_per_msm_two_in_flight'], ex['C3_ipa_prover']['value'], ex['C3_ipa_prover']['with_fixed_generators']['seconds'], ex['C4_aggregated_range_proof']['value'], c5['value'], c5['wire_format_2']['value'], c5['batch_latency_s']*1e3, c5['wire_format_2']['batch_latency_s']*1e3, c5['batch_prover']['proves_per_s']))"
done
