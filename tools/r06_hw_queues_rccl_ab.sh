for r in 1 2; do for q in 8 16; do
BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 GPU_MAX_HW_QUEUES=$q python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2954$r bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --soak-seconds 0 2>/dev/null | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); s=d['summary']
print('round $r  RCCL initialised, GPU_MAX_HW_QUEUES=$q  step %.4f ms (exchange %s us) | C2 %.4f / %.4f / %.4f | C3 %.5f | C5 %.4g / v2 %s / v3 %s' % (s['ms_per_step'], d.get('exchange_us'), s['C2_ms_one'], s['C2_ms_two'], s['C2_ms_three'], s['C3_s'], s['C5_verifies_per_s'], s['C5_v2_verifies_per_s'], s['C5_v3_verifies_per_s']))"
done; done
