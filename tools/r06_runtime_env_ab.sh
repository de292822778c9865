#!/bin/bash
# A/B on one box of HIP-runtime environment knobs (read when HIP initialises) over the whole default bench.
#   ENVS="HIP_FORCE_DEV_KERNARG=0 HIP_FORCE_DEV_KERNARG=1 ROC_ACTIVE_WAIT_TIMEOUT=100" ROUNDS=2 bash tools/r06_runtime_env_ab.sh
for r in $(seq 1 ${ROUNDS:-2}); do for e in default ${ENVS:-HIP_FORCE_DEV_KERNARG=0 HIP_FORCE_DEV_KERNARG=1}; do
  if [ "$e" = default ]; then pre=""; else pre="$e"; fi
  env $pre python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --soak-seconds 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); s=d['summary']
print('round $r  %-32s step %.4f | C2 %.4f / %.4f / %.4f | C3 %.5f | C4 %.5f / %.5f | C5 %.4g / %.4g / %.4g, one %.3f | prover %.4g' % ('$e', s['ms_per_step'], s['C2_ms_one'], s['C2_ms_two'], s['C2_ms_three'], s['C3_s'], s['C4_prove_s'], s['C4_verify_s'], s['C5_verifies_per_s'], s['C5_v2_verifies_per_s'], s['C5_v3_verifies_per_s'], s['C5_one_batch_ms'], s['prover_proofs_per_s']))"
done; done
