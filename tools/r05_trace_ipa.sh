#!/bin/bash
# kernel + copy timeline of the inner-product prover's rounds at n = 2^13 (the shape of config C4's argument): what a late round's 0.21 ms is made of
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_trace_ipa
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/tl -o tl -- python3 $R/tools/c3_round_times.py ${1:-13} > $OUT/run.txt 2>&1
tail -18 $OUT/run.txt
DB=$(find $OUT/tl -name "*results.db" | head -1)
NROWS=$(python3 $R/tools/rocpd_timeline.py $DB timeline 0 100000 | wc -l)
python3 $R/tools/rocpd_timeline.py $DB timeline $((NROWS - 60)) 60
rm -rf $OUT/tl
