#!/bin/bash
# One gpurun call's worth of profiling evidence for profiles/ (run from the repo root ON THE GPU BOX):
#   bash tools/profile_round.sh r02p
# rocprofv3 per-kernel stats of the bench, the two --pmc passes behind roofline.traffic, per-kernel stats of the C3 and
# C5 configurations, and the clocks rocm-smi reports before / after.  Counter passes are separate runs with
# --kernel-trace only (no --stats / sys-trace next to --pmc).
set -u
TAG=${1:-r03p}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocm-smi --showclocks --showpower > $OUT/rocm_smi_before.txt 2>&1
B="python3 $R/bench.py --no-cpu-baseline --no-extra --soak-seconds 0"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bench -- $B --steps 20 --warmup 3 --preheat-ms 60 > $OUT/bench_under_rocprofv3.json 2> $OUT/stats_bench.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_FETCH_SIZE -- $B --steps 3 --warmup 1 --preheat-ms 0 > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_WRITE_SIZE -- $B --steps 3 --warmup 1 --preheat-ms 0 > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3 -- python3 $R/tools/bench_ipa_sharded.py 20 > $OUT/c3.json 2> $OUT/stats_c3.err
export C5_PINNED=1 C5_PREPARE=device C5_ONECALL=1          # the batch verifier's one-call path on a page-locked receive buffer
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5 -- python3 $R/tools/profile_c5.py > $OUT/c5.txt 2> $OUT/stats_c5.err
export C5_SERIAL=1                                         # ... and with the point decoding BEHIND the preparation kernels: every kernel's own duration
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5_serial -- python3 $R/tools/profile_c5.py > $OUT/c5_serial.txt 2> $OUT/stats_c5_serial.err
unset C5_SERIAL
rocm-smi --showclocks --showpower > $OUT/rocm_smi_after.txt 2>&1
cd $R
python3 tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE > $OUT/pmc_traffic.json 2> $OUT/pmc_traffic.err
for d in stats_bench stats_c3 stats_c5 stats_c5_serial; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/${d}_kernel_stats.csv; done
# keep the merge small: the raw traces are large
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
ls -la $OUT
