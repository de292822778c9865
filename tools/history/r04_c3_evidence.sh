#!/bin/bash
# Two pieces of C3 evidence: (1) the kernel timeline of the rounds behind the 16-way fold (pairs of 65 537-pair MSMs on two lanes),
# (2) the product fold with shared GLV halves against per-lane products (option fold_shared), per-round table.
#   bash tools/r04_c3_evidence.sh <outdir>
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/${1:-gpurun_out/r04c3}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/tl -o tl -- python3 $R/tools/c3_round_times.py 20 > $OUT/rounds_under_rocprof.txt 2> $OUT/tl.err
DB=$(find $OUT/tl -name "*results.db" | head -1)
python3 $R/tools/rocpd_timeline.py $DB after k_ec_multifold 64 > $OUT/timeline_mid_round.txt
rm -rf $OUT/tl
cd $R
for o in "fold_shared=1" "fold_shared=0" "fold_shared=1" "fold_shared=0"; do echo "== $o"; timeout 200 python tools/c3_round_times.py 20 $o 2>&1 | grep -E " 8192 |total"; done > $OUT/fold_shared_ab.txt
cat $OUT/fold_shared_ab.txt; head -50 $OUT/timeline_mid_round.txt
