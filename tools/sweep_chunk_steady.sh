#!/bin/bash
# chunk length L of the accumulate kernel x the two-lane pipeline at n = 2^20, at steady clocks (bench.py's preheat), one call
B="python3 bench.py --no-extra --no-cpu-baseline --soak-seconds 0 --steps 200 --warmup 10"
for L in ${LS:-64 76 86 96 108 128}; do
  timeout 300 $B --opt chunk=$L > gpurun_out/bench_s.json 2>/dev/null
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/bench_s.json').read().strip().split('\n')[-1])
print('L=%s  ms_per_step %.4f  pairs/s %.4g  accumulate in the pipeline %.4f  alone %.4f  ok %s' % (sys.argv[1], d['ms_per_step'], d['value'], d['roofline']['kernel_avg_ms'], d['stage_ms_per_msm']['msm_accumulate'], d['result_ok']))" $L
done
