#!/bin/bash
# What do the two HIP events per step around k_accum_l0 (bench.py's live kernel timing) cost the pipelined step, and how much
# do the fill and the drain of the two-deep pipeline weigh at 20 steps?  Run on the GPU box from the repo root.
B="python3 bench.py --no-extra --no-cpu-baseline --soak-seconds 0"
show() { python3 -c "
import json,sys
d=json.loads(open('gpurun_out/bench_k.json').read().strip().split('\n')[-1])
print(sys.argv[1], 'ms_per_step', round(d['ms_per_step'],4), 'hip_event', round(d['hip_event_ms_per_step'],4), 'pairs/s %.4g' % d['value'])" "$1"; }
for i in 1 2; do
  BENCH_NO_KERNEL_EVENTS= timeout 300 $B --steps 20 > gpurun_out/bench_k.json 2>/dev/null; show "kernel events ON , 20 steps:"
  BENCH_NO_KERNEL_EVENTS=1 timeout 300 $B --steps 20 > gpurun_out/bench_k.json 2>/dev/null; show "kernel events OFF, 20 steps:"
done
BENCH_NO_KERNEL_EVENTS= timeout 300 $B --steps 100 --warmup 5 > gpurun_out/bench_k.json 2>/dev/null; show "kernel events ON , 100 steps:"
BENCH_NO_KERNEL_EVENTS=1 timeout 300 $B --steps 100 --warmup 5 > gpurun_out/bench_k.json 2>/dev/null; show "kernel events OFF, 100 steps:"
