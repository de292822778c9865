#!/bin/bash
# kernel timeline of ONE batch verification at a time (2^14 proofs, wire format 2, one native call per batch)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_trace_c5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export C5_WIRE=${C5_WIRE:-2} C5_PINNED=1 C5_ONECALL=1 C5_NO_STAGE_TIMERS=1 C5_REPS=6 C5_DISTINCT=64
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/tl -o tl -- python3 $R/tools/profile_c5.py > $OUT/run.txt 2>&1
grep "rep " $OUT/run.txt
DB=$(find $OUT/tl -name "*results.db" | head -1)
NROWS=$(python3 $R/tools/rocpd_timeline.py $DB timeline 0 100000 | wc -l)
python3 $R/tools/rocpd_timeline.py $DB timeline $((NROWS - 40)) 40
rm -rf $OUT/tl
