#!/bin/bash
# chunk length of the chained two-lane pipeline with the fused wave scan (round 4), one box, one call
out=${1:-gpurun_out/r04_chunk}
mkdir -p "$out"
for L in 86 64 72 80 96 104 112 128 86 96; do
  line=$(python bench.py --steps 200 --warmup 10 --no-extra --no-cpu-baseline --soak-seconds 0 --opt chunk=$L 2>>"$out/stderr.txt" | tail -1)
  python - "$L" "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print("L=%-4s ms_per_step %.4f  pairs/s %.4g  accum in the pipeline %.4f  alone %.4f  ok %s" % (sys.argv[1], d["ms_per_step"], d["value"], d["roofline"]["kernel_avg_ms"], d["stage_ms_per_msm"]["msm_accumulate"], d["result_ok"]))
PY
done | tee "$out/chunk_sweep.txt"
