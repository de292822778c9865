#!/bin/bash
# kernel timelines of synchronous 2^16-pair MSMs under a few option sets:  bash tools/r05_trace_mid.sh <tag> "opt=v opt=v" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_trace_mid
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace -d $OUT/tl$i -o tl -- python3 $R/tools/r05_msm_loop.py ${N:-65536} 60 $cfg > $OUT/run$i.txt 2>&1
  DB=$(find $OUT/tl$i -name "*results.db" | head -1)
  echo "#### config $i: $cfg" > $OUT/cfg$i.txt
  tail -1 $OUT/run$i.txt >> $OUT/cfg$i.txt
  python3 $R/tools/rocpd_timeline.py $DB stats >> $OUT/cfg$i.txt
  NROWS=$(python3 $R/tools/rocpd_timeline.py $DB timeline 0 100000 | wc -l)
  python3 $R/tools/rocpd_timeline.py $DB timeline $((NROWS - 26)) 26 >> $OUT/cfg$i.txt
  rm -rf $OUT/tl$i
  cat $OUT/cfg$i.txt
done
