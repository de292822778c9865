#!/bin/bash
# kernel timeline of the two-lane pipeline (rocprofv3 --kernel-trace) for a chunk length, without bench.py's own events
L=${1:-86}
OUT=gpurun_out/trace_L$L
cd /tmp && export TMPDIR=/tmp
BENCH_NO_KERNEL_EVENTS=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT -- python3 $GRAFT_REPO_ROOT/bench.py --no-extra --no-cpu-baseline --soak-seconds 0 --steps 12 --warmup 4 --preheat-ms 80 --opt chunk=$L > $GRAFT_REPO_ROOT/$OUT.json 2> $GRAFT_REPO_ROOT/$OUT.err
cd $GRAFT_REPO_ROOT
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 tools/trace_timeline.py "$f" > $OUT.timeline.txt
find $OUT -name "*.csv" -size +1M -delete
