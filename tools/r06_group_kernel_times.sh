#!/bin/bash
# The headline loop on one GPU without and with an initialised RCCL process group (one rank, NO exchange per step): step time and the
# per-stage kernel times (HIP events on the launch streams).
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29550 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
run() { python3 bench.py --gpus 1 --steps 40 --warmup 6 --no-cpu-baseline --soak-seconds 0 --no-extra 2>/dev/null | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); r=d['roofline']
print('%-18s ms_per_step %.4f  hip_event_ms_per_step %s  accumulate in the pipeline %s ms  stages %s' % (sys.argv[1], d['ms_per_step'], d.get('hip_event_ms_per_step'), r.get('kernel_avg_ms'), d.get('stage_ms_per_msm')))" "$1"; }
unset BENCH_FORCE_DIST; run plain
export BENCH_FORCE_DIST=1 BENCH_EXCHANGE=0; run rccl_group_only
