#!/bin/bash
# What a process group costs the headline loop on ONE GPU (a group of one rank over RCCL, no torchrun): plain, with the group but WITHOUT the
# per-step exchange (BENCH_EXCHANGE=0), and with it; the driver's 20 steps and 400 steps; host milliseconds per step by phase.
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29547 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for steps in 20 400; do
for mode in plain group_only dist; do
  unset BENCH_FORCE_DIST BENCH_EXCHANGE
  [ $mode != plain ] && export BENCH_FORCE_DIST=1
  [ $mode = group_only ] && export BENCH_EXCHANGE=0
  python3 bench.py --gpus 1 --steps $steps --warmup 5 --no-cpu-baseline --soak-seconds 0 --no-extra 2>/dev/null | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1])
print('%4d steps  %-10s ms_per_step %.4f  host ms per step %s  exchange_us %s' % ($steps, '$mode', d['ms_per_step'], d.get('host_ms_per_step'), d.get('exchange_us')))"
done
done
