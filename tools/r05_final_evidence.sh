#!/bin/bash
# Round 5's closing evidence in ONE gpurun call (from the repo root on the GPU box):  bash tools/r05_final_evidence.sh [outdir]
out=${1:-gpurun_out/r05z}
mkdir -p "$out"
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -12 > "$out/pytest_gpu.txt"; tail -3 "$out/pytest_gpu.txt"
for k in 1 2 3; do timeout 900 python bench.py --steps 20 --warmup 5 > "$out/bench_$k.json" 2> "$out/bench_$k.err"; done
timeout 600 python bench.py --steps 200 --warmup 10 --no-extra --no-cpu-baseline > "$out/bench_200steps.json" 2> "$out/bench_200steps.err"
python3 - "$out" <<'PY'
import json, sys
out = sys.argv[1]
rows = []
for k in (1, 2, 3):
    try:
        d = json.loads(open("%s/bench_%d.json" % (out, k)).read().strip().splitlines()[-1])
    except Exception as e:
        rows.append("run %d: unreadable (%s)" % (k, e)); continue
    ex = d["extra"]
    c5, c3, c2 = ex["C5_batch_verify"], ex["C3_ipa_prover"], ex["C2_msm_2e16"]
    rows.append("run %d (--steps 20 --warmup 5): %.4f ms/step  %.4g pairs/s  ok %s | C2 %.4f one at a time %.4f two in flight %.4f three | C3 %.4f s (fixed generators %s) | C4 prove %.5f s | C5 %.4g verifies/s (v2 %.4g; one at a time %.3f / %.3f ms) | batch prover %.4g proofs/s (device %.4g)" % (
        k, d["ms_per_step"], d["value"], d["result_ok"], c2.get("ms_per_msm_one_at_a_time", -1), c2.get("ms_per_msm_two_in_flight", -1), c2.get("ms_per_msm_three_in_flight", -1), c3["value"],
        (c3.get("with_fixed_generators") or {}).get("seconds"), ex["C4_aggregated_range_proof"]["value"], c5["value"], c5["wire_format_2"]["value"],
        c5.get("batch_latency_s", -1) * 1e3, c5["wire_format_2"].get("batch_latency_s", -1) * 1e3, c5["batch_prover"]["proves_per_s"], c5["batch_prover"]["proves_per_s_device_time"]))
try:
    d = json.loads(open(out + "/bench_200steps.json").read().strip().splitlines()[-1])
    rows.append("--steps 200 --warmup 10 (headline only): %.4f ms/step  %.4g pairs/s  ok %s" % (d["ms_per_step"], d["value"], d["result_ok"]))
except Exception as e:
    rows.append("200-step run unreadable (%s)" % e)
open(out + "/bench_three_runs.txt", "w").write("\n".join(rows) + "\n")
print("\n".join(rows))
PY
timeout 300 python tools/bench_prove_batch.py 8 10 12 14 16 > "$out/prove_bench.txt" 2>&1; tail -6 "$out/prove_bench.txt"
timeout 500 python tools/fuzz_msm.py 300 > "$out/fuzz_msm.txt" 2>&1; tail -2 "$out/fuzz_msm.txt"
timeout 300 python tools/fuzz_ops.py 100 > "$out/fuzz_ops.txt" 2>&1; tail -2 "$out/fuzz_ops.txt"
timeout 300 python tools/fuzz_batch_prepare.py 100 > "$out/fuzz_batch_prepare.txt" 2>&1; tail -2 "$out/fuzz_batch_prepare.txt"
timeout 1500 bash tools/profile_round.sh "$(basename $out)p" > "$out/profile_round.log" 2>&1
R=${GRAFT_REPO_ROOT:-$(pwd)}
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats_prove -- python3 $R/tools/bench_prove_batch.py 14 > $R/$out/prove_under_rocprofv3.txt 2> $R/$out/stats_prove.err )
f=$(find $out/stats_prove -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/stats_prove_kernel_stats.csv
find $out -name "*kernel_trace.csv" -size +2M -delete
python tools/build_report.py > "$out/build_resource_table.txt" 2>&1 || true
ls $out | head -40
