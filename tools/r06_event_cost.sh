#!/bin/bash
# What the two HIP events per step around the dominant kernel (the live roofline measurement) cost the driver's 20-step region.
for r in 1 2 3; do for e in 0 1; do
  if [ $e = 1 ]; then export BENCH_NO_KERNEL_EVENTS=1; else unset BENCH_NO_KERNEL_EVENTS; fi
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --soak-seconds 0 --no-extra 2>/dev/null | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1])
print('round $r  kernel events %s  ms_per_step %.4f' % ('off' if $e else 'on ', d['ms_per_step']))"
done; done
