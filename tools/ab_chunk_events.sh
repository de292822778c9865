#!/bin/bash
# chunk length of the two-lane pipeline x the live kernel events of bench.py (do the events change which L wins?)
B="python3 bench.py --no-extra --no-cpu-baseline --soak-seconds 0 --steps 200 --warmup 10"
for rep in 1 2; do for ev in "" 1; do for L in 86 128; do
  BENCH_NO_KERNEL_EVENTS=$ev timeout 300 $B --opt chunk=$L > gpurun_out/bench_s.json 2>/dev/null
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/bench_s.json').read().strip().split('\n')[-1])
print('events %-3s L=%-3s ms_per_step %.4f' % ('off' if sys.argv[2] else 'on', sys.argv[1], d['ms_per_step']))" $L "$ev"
done; done; done
