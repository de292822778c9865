#!/bin/bash
# Round 4's closing evidence in ONE gpurun call (from the repo root on the GPU box):  bash tools/r04_final_evidence.sh [outdir]
#   the whole GPU suite, the default bench three times, the per-round table of the C3 prover, the fuzzers, the profiling round
out=${1:-gpurun_out/r04z}
mkdir -p "$out"
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -12 > "$out/pytest_gpu.txt"; tail -3 "$out/pytest_gpu.txt"
for k in 1 2 3; do timeout 600 python bench.py > "$out/bench_$k.json" 2> "$out/bench_$k.err"; done
python3 - "$out" <<'PY'
import json, sys
out = sys.argv[1]
rows = []
for k in (1, 2, 3):
    try:
        d = json.load(open("%s/bench_%d.json" % (out, k)))
    except Exception as e:
        rows.append("run %d: unreadable (%s)" % (k, e)); continue
    ex = d["extra"]
    c5 = ex["C5_batch_verify"]
    rows.append("run %d: %.4f ms/step  %.4g pairs/s  ok %s | C2 %.3f ms | C3 %.4f s | C4 prove %.5f verify %.5f s | C5 %.4g verifies/s (v2 %.4g; one at a time %.3f / %.3f ms)" % (
        k, d["ms_per_step"], d["value"], d["result_ok"], ex["C2_msm_2e16"].get("ms_per_msm_one_at_a_time", -1), ex["C3_ipa_prover"]["value"],
        ex["C4_aggregated_range_proof"]["prove_s"], ex["C4_aggregated_range_proof"]["verify_s"], c5["value"], c5["wire_format_2"]["value"],
        c5.get("batch_latency_s", -1) * 1e3, c5["wire_format_2"].get("batch_latency_s", -1) * 1e3))
open(out + "/bench_three_runs.txt", "w").write("\n".join(rows) + "\n")
print("\n".join(rows))
PY
for o in "fold_wnaf=2" "fold_wnaf=1" "fold_wnaf=2"; do echo "== $o"; timeout 200 python tools/c3_round_times.py 20 $o 2>&1 | tail -24; done > "$out/c3_rounds.txt"; grep -E "==|total" "$out/c3_rounds.txt"
timeout 400 python tools/fuzz_msm.py 150 > "$out/fuzz_msm.txt" 2>&1; tail -2 "$out/fuzz_msm.txt"
timeout 400 python tools/fuzz_ops.py 150 > "$out/fuzz_ops.txt" 2>&1; tail -2 "$out/fuzz_ops.txt"
timeout 400 python tools/fuzz_batch_prepare.py 100 > "$out/fuzz_batch_prepare.txt" 2>&1; tail -2 "$out/fuzz_batch_prepare.txt"
timeout 1500 bash tools/profile_round.sh "$(basename $out)p" > "$out/profile_round.log" 2>&1
