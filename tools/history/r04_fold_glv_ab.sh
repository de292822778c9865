#!/bin/bash
# rocprofv3 kernel stats of the C3 prover's 16-way generator fold, fold_wnaf = 1 (whole coefficients) against 2 (GLV halves):
#   bash tools/r04_fold_glv_ab.sh <outdir>     (on the GPU box, from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/${1:-gpurun_out/r04f}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in 1 2; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fold$m -- python3 $R/tools/c3_round_times.py 20 fold_wnaf=$m > $OUT/fold$m.txt 2> $OUT/fold$m.err
  f=$(find $OUT/fold$m -name "*kernel_stats.csv" | head -1)
  echo "== fold_wnaf=$m"; grep -E "total|131072" $OUT/fold$m.txt
  grep -E "multifold|odd_multiples" "$f" | cut -d, -f1-4 | sed 's/(.*)"/"/'
  find $OUT/fold$m -name "*kernel_trace.csv" -delete
done
