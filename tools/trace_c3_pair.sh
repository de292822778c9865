#!/bin/bash
# kernel timeline of the large rounds of the inner-product prover (two MSMs of 2^20 pairs per round), pair_schedule = $1
PS=${1:-1}
OUT=gpurun_out/trace_c3_ps$PS
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT -- python3 $GRAFT_REPO_ROOT/tools/c3_round_times.py 20 pair_schedule=$PS > $GRAFT_REPO_ROOT/$OUT.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 tools/trace_timeline.py "$f" 500 8 > $OUT.timeline.txt
find $OUT -name "*.csv" -size +1M -delete
