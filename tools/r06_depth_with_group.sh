export MASTER_ADDR=127.0.0.1 MASTER_PORT=29549 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
run() { python3 bench.py --gpus 1 --steps 40 --warmup 6 --no-cpu-baseline --soak-seconds 0 --no-extra $2 2>/dev/null | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1])
print('%-40s ms_per_step %.4f  host %s' % (sys.argv[1], d['ms_per_step'], d.get('host_ms_per_step')))" "$1"; }
for r in 1 2; do
unset BENCH_FORCE_DIST
run "plain, depth 2" "--depth 2"
run "plain, depth 3" "--depth 3"
export BENCH_FORCE_DIST=1
run "rccl group + exchange, depth 2" "--depth 2"
run "rccl group + exchange, depth 3" "--depth 3"
done
