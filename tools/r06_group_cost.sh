export MASTER_ADDR=127.0.0.1 MASTER_PORT=29548 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 BENCH_FORCE_DIST=1 BENCH_EXCHANGE=0
run() { python3 bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --soak-seconds 0 --no-extra 2>/dev/null | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1])
print('%-52s ms_per_step %.4f' % (sys.argv[1], d['ms_per_step']))" "$1"; }
for r in 1 2; do
run "rccl group, 16 queues"
BENCH_DIST_BACKEND=gloo run "gloo group, 16 queues"
GPU_MAX_HW_QUEUES=8 run "rccl group, 8 queues"
GPU_MAX_HW_QUEUES=32 run "rccl group, 32 queues"
TORCH_NCCL_HIGH_PRIORITY=0 run "rccl group, 16 queues, TORCH_NCCL_HIGH_PRIORITY=0"
done
