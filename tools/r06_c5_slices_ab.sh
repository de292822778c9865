#!/bin/bash
# A/B on one box: uploads per batch of the batch verifier (option rp_slices; BPMI_OPTIONS reaches every engine of the bench's batches in
# flight), wire format 3 and 2: one batch alone (tools/profile_c5.py, one native call) and eight in flight (the bench's C5 extra).
for r in 1 2; do
for s in 1 4 2; do
  export BPMI_OPTIONS="rp_slices=$s"
  for w in 3 2; do
    lat=$(C5_REPS=10 C5_NO_STAGE_TIMERS=1 C5_WIRE=$w C5_ONECALL=1 C5_PINNED=1 C5_PREPARE=device python3 tools/profile_c5.py 2>&1 | grep "^prepare" | tail -6 | awk '{print $5}' | sort -n | head -3 | tr '\n' ' ')
    echo "round $r  rp_slices=$s  format $w  one batch alone (three best of the last six, ms): $lat"
  done
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --soak-seconds 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); c=d['extra']['C5_batch_verify']
print('round $r  rp_slices=$s  eight in flight: format 1 %.4g  format 2 %.4g  format 3 %.4g verifies/s | one batch: %.3f / %.3f / %.3f ms' % (c['value'], c['wire_format_2']['value'], c['wire_format_3']['value'], c['batch_latency_s']*1e3, c['wire_format_2']['batch_latency_s']*1e3, c['wire_format_3']['batch_latency_s']*1e3))"
done
done
