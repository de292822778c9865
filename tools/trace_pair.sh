#!/bin/bash
# kernel timeline of two MSMs that are needed together (tools/pair_modes.py, chained lanes)
OUT=gpurun_out/trace_pair
cd /tmp && export TMPDIR=/tmp
PAIR_ONLY="lanes chained" timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT -- python3 $GRAFT_REPO_ROOT/tools/pair_modes.py 20 > $GRAFT_REPO_ROOT/$OUT.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 tools/trace_timeline.py "$f" > $OUT.timeline.txt
find $OUT -name "*.csv" -size +1M -delete
