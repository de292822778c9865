#!/bin/bash
# A/B on one box: the HIP runtime's hardware-queue count (GPU_MAX_HW_QUEUES, read when HIP initialises; benchlib/common.py sets a default) for the
# whole default bench: the headline step, C2, C3, the batch verifier with eight batches in flight (each with an engine = two or three streams
# of its own), the batched prover.   QS="8 12 16 20" ROUNDS=3 bash tools/r06_hw_queues_ab.sh
for r in $(seq 1 ${ROUNDS:-2}); do for q in ${QS:-8 16 24}; do
GPU_MAX_HW_QUEUES=$q python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --soak-seconds 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); c=d['extra']['C5_batch_verify']; s=d['summary']
print('round $r  GPU_MAX_HW_QUEUES=$q  step %.4f ms | C2 %.4f / %.4f / %.4f | C3 %.5f | C5 format 1 %.4g  2 %.4g  3 %.4g verifies/s | prover %.4g' % (s['ms_per_step'], s['C2_ms_one'], s['C2_ms_two'], s['C2_ms_three'], s['C3_s'], c['value'], c['wire_format_2']['value'], c['wire_format_3']['value'], s['prover_proofs_per_s']))"
done; done
