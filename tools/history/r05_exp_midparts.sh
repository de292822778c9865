#!/bin/bash
# k_msm_mid with a window's pairs over 1 .. 4 blocks: the inner-product prover at n = 2^13 (C4's argument) and 2^20 (C3), then the bench
for parts in 1 2 3 4 0; do
  echo "mid_parts=$parts logn=13 $(timeout 120 python tools/c3_round_times.py 13 mid_parts=$parts 2>&1 | tail -1)"
done
for parts in 1 3 0; do
  echo "mid_parts=$parts logn=20 $(timeout 300 python tools/c3_round_times.py 20 mid_parts=$parts 2>&1 | tail -1)"
done
for cfg in "--opt mid_parts=1" "" "--opt mid_parts=1" ""; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $cfg > gpurun_out/_b.json 2>/dev/null
  python3 -c "
import json
d=json.loads(open('gpurun_out/_b.json').read().strip().splitlines()[-1]); ex=d['extra']; c5=ex['C5_batch_verify']
print('cfg [$cfg]: %.4f ms/step | C3 %.5f (fixed %.5f) | C4 prove %.5f verify %s | C5 %.4g' % (d['ms_per_step'], ex['C3_ipa_prover']['value'], ex['C3_ipa_prover']['with_fixed_generators']['seconds'], ex['C4_aggregated_range_proof']['value'], ex['C4_aggregated_range_proof'].get('verify_seconds'), c5['value']))"
done
