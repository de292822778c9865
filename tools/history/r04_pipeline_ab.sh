#!/bin/bash
# Round 4: A/B of the pipeline variants on ONE box in ONE call (box-to-box spread is larger than the effects):
# fused wave scan on / off, two / three MSMs in flight, chunk lengths at the wave-quantisation points.
#   bash tools/r04_pipeline_ab.sh OUTDIR
out=${1:-gpurun_out/r04_ab}
mkdir -p "$out"
run() {   # label, extra args
  label=$1; shift
  line=$(python bench.py --steps 200 --warmup 10 --no-extra --no-cpu-baseline --soak-seconds 0 "$@" 2>>"$out/stderr.txt" | tail -1)
  echo "$line" > "$out/bench_$label.json"
  python - "$label" "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print("%-34s ms_per_step %.4f  pairs/s %.4g  accum_avg_ms %.4f  ok %s" % (sys.argv[1], d["ms_per_step"], d["value"], d["roofline"]["kernel_avg_ms"], d["result_ok"]))
PY
}
for rep in 1 2; do
  run "r03path_unfused_d2_$rep" --opt fused_scan=0
  run "fused_d2_$rep"
  run "fused_d3_$rep" --depth 3
  run "fused_d3_L128_$rep" --depth 3 --opt chunk=128
  run "fused_d2_L128_$rep" --opt chunk=128
  run "fused_d3_L64_$rep" --depth 3 --opt chunk=64
done | tee "$out/summary.txt"
