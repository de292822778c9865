#!/bin/bash
# The headline pipeline's kernel timeline: rocprofv3 --kernel-trace of a short bench run, three steady-state steps printed.
#   bash tools/bench_timeline.sh <outdir> [extra bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/${1:-gpurun_out/r04t}
shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/tl -o tl -- python3 $R/bench.py --no-cpu-baseline --no-extra --soak-seconds 0 --steps 24 --warmup 4 --preheat-ms 60 "$@" > $OUT/bench.json 2> $OUT/bench.err
DB=$(find $OUT/tl -name "*results.db" | head -1)
python3 $R/tools/rocpd_timeline.py $DB stats > $OUT/stats.txt
N=$(python3 $R/tools/rocpd_timeline.py $DB timeline 0 100000 | wc -l)
python3 $R/tools/rocpd_timeline.py $DB timeline $((N / 2)) 64 > $OUT/timeline.txt
rm -rf $OUT/tl
head -c 300 $OUT/bench.json; echo; cat $OUT/timeline.txt
