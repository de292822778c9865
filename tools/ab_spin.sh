#!/bin/bash
# polling before sleeping in the library's waits (option spin_wait): config C5 with eight batches in flight, C2 one at a time, C4
for rep in 1 2; do for sp in 0 30000; do
  timeout 700 python3 bench.py --no-cpu-baseline --soak-seconds 0 --opt spin_wait=$sp > gpurun_out/bench_sp.json 2>/dev/null
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/bench_sp.json').read().strip().split('\n')[-1]); e=d['extra']; c=e['C5_batch_verify']
print('spin_wait=%-6s C5 %.4g verifies/s (one at a time %.3f ms) | C2 one at a time %.4f ms | C3 %.5f s | C4 %.5f / %.5f s' % (sys.argv[1], c['value'], c['batch_latency_s']*1e3, e['C2_msm_2e16']['ms_per_msm_one_at_a_time'], e['C3_ipa_prover']['value'], e['C4_aggregated_range_proof']['prove_s'], e['C4_aggregated_range_proof']['verify_s']))" $sp
done; done
