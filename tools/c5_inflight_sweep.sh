#!/bin/bash
# verifies/s of the batch verifier against the number of batches in flight (one host thread + engine + receive buffer each)
for k in ${KS:-4 6 8 10 12 16}; do
  BENCH_C5_INFLIGHT=$k timeout 600 python3 bench.py --no-cpu-baseline --soak-seconds 0 --steps 20 --warmup 3 > gpurun_out/bench_c5s.json 2>/dev/null
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/bench_c5s.json').read().strip().split('\n')[-1])
c=d['extra']['C5_batch_verify']
g=lambda f: (c.get('wire_format_%d' % f) or {}).get('value') or 0.0
print('in flight %2s: format 1 %.4g verifies/s (%.4f ms per batch, one at a time %.3f ms) | format 2 %.4g | format 3 %.4g' % (sys.argv[1], c['value'], c['seconds_per_batch']*1e3, c['batch_latency_s']*1e3, g(2), g(3)))" $k
done
